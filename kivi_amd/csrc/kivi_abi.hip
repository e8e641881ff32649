// Error plumbing and version of the C ABI (include/kivi_hip.h).
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "kivi_common.h"

static thread_local char g_err[512] = "";

void kivi_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

extern "C" int kivi_abi_version(void) { return KIVI_ABI_VERSION; }
extern "C" const char* kivi_last_error(void) { return g_err; }

// ---- sticky device-side error (include/kivi_hip.h, kivi_device_error): two ints of pinned, host-coherent memory -- [0] the error
// code a kernel left (0 = none), [1] the unit it was for -- that the kernels of a sliced launch get a device pointer to.  The only
// allocation the library makes; if it fails the pointer stays null and a timeout is visible as NaN (and in the workspace's error
// word) only.
#include <mutex>
static int* g_dev_err_host = nullptr;
static int* g_dev_err_dev = nullptr;
int* kivi_device_error_word(hipStream_t stream) {
    // (never allocate while `stream` is being captured into a graph: an allocation call is not a capturable operation and would
    // invalidate the capture.  A launch captured before the word exists simply carries a null pointer: its timeout would show as NaN
    // and in the workspace's error word only.  GraphedDecode runs the first step of a geometry class eagerly, which allocates it.)
    if (!g_dev_err_dev) {
        hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(stream, &st) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        if (st != hipStreamCaptureStatusNone) return nullptr;
    }
    static std::once_flag once;
    std::call_once(once, [] {
        void* h = nullptr;
        if (hipHostMalloc(&h, 64, hipHostMallocMapped | hipHostMallocPortable) != hipSuccess || !h) { (void)hipGetLastError(); return; }
        memset(h, 0, 64);
        void* d = nullptr;
        if (hipHostGetDevicePointer(&d, h, 0) != hipSuccess || !d) { (void)hipGetLastError(); (void)hipHostFree(h); return; }
        g_dev_err_host = (int*)h;
        g_dev_err_dev = (int*)d;
    });
    return g_dev_err_dev;
}
// returns and clears the error; `unit` (or null) receives the unit it was recorded for
int kivi_take_device_error(int* unit) {
    if (!g_dev_err_host) return 0;
    const int e = __atomic_exchange_n(g_dev_err_host, 0, __ATOMIC_ACQ_REL);
    if (e && unit) *unit = __atomic_load_n(g_dev_err_host + 1, __ATOMIC_ACQUIRE);
    return e;
}
extern "C" int kivi_device_error(void) {
    int unit = -1;
    const int e = kivi_take_device_error(&unit);
    if (e) kivi_set_error("a block of an earlier sliced decode launch gave up waiting for a partner block of (batch row, kv head) unit %d: "
                          "that step's output holds NaN for the unit", unit);
    return e ? KIVI_ETIMEOUT : 0;
}

// ---- per-dispatch timing events (instrumentation for bench.py / tools; not on the drop-in surface)
static thread_local KiviLaunchEvents g_events = {nullptr, nullptr};

KiviLaunchEvents kivi_take_launch_events() {
    KiviLaunchEvents e = g_events;
    g_events.start = nullptr;
    g_events.stop = nullptr;
    return e;
}

static thread_local char g_timed[160] = "";
void kivi_note_timed_kernel(const char* name) { snprintf(g_timed, sizeof g_timed, "%s", name); }
extern "C" const char* kivi_last_timed_kernel(void) { return g_timed; }

extern "C" void* kivi_event_create(void) {
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    return (void*)e;
}
extern "C" void kivi_event_destroy(void* e) {
    if (e) (void)hipEventDestroy((hipEvent_t)e);
}
extern "C" void kivi_set_launch_events(void* start, void* stop) {
    g_events.start = (hipEvent_t)start;
    g_events.stop = (hipEvent_t)stop;
}
extern "C" float kivi_event_elapsed_us(void* start, void* stop) {
    float ms = -1.0f;
    if (hipEventSynchronize((hipEvent_t)stop) != hipSuccess) return -1.0f;
    if (hipEventElapsedTime(&ms, (hipEvent_t)start, (hipEvent_t)stop) != hipSuccess) return -1.0f;
    return ms * 1000.0f;
}

// ---- phase time stamps of the fused row kernels (tools/row_phases.py, tools/mf_row_phases.py): a caller-owned device buffer of
// grid x 4 waves x 16 uint64.  Only -DKIVI_TUNING builds carry the stamping instantiations and remember the pointer; in the product
// library the call is accepted and does nothing (no process-global state, no diagnostic kernels).
#ifdef KIVI_TUNING
static unsigned long long* g_stamps = nullptr;
unsigned long long* kivi_debug_stamps() { return g_stamps; }
extern "C" void kivi_debug_set_stamps(void* buf) { g_stamps = (unsigned long long*)buf; }
#else
unsigned long long* kivi_debug_stamps() { return nullptr; }
extern "C" void kivi_debug_set_stamps(void*) {}
#endif
