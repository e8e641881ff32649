"""Time the literal pybind twin (kivi_gemv.gemv_forward_cuda_outer_dim -> kivi_gemv_outer_dim, the reference's own kernel-input layout:
what an unmodified quant/matmul.py:198-219 calls after its three transposes) at BASELINE configs[1] (C2: B=32, H=32, T=4096, D=128,
g=32, 2-bit) for qK^T and at the matching sV shape, against the HBM roofline.  HIP events on torch's stream around every call, 12
rotating caches (12 x 200 MiB >> the 256 MiB Infinity Cache).  Also times the reference wrapper's transposes (matmul.py:205,213-214)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from kivi_amd.quant import kivi_gemv, matmul, new_pack

B, nh, T, D, g = 32, 32, 4096, 128, 32
bits = int(os.environ.get("BITS", "2"))
NC = 12


def timed(fn, n):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for i in range(n):
        ev[i][0].record(); fn(i); ev[i][1].record()
    torch.cuda.synchronize()
    us = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
    return us[len(us) // 2], us[0], sum(us) / len(us)


def report(name, nbytes, fn):
    for i in range(NC):
        fn(i)
    med, mn, avg = timed(lambda i: fn(i % NC), 10 * NC)
    print(f"{name}: median {med:.1f} us  min {mn:.1f}  avg {avg:.1f}  algorithmic {nbytes / 1e6:.1f} MB  ->  {nbytes / (med * 1e-6) / 8e12:.3f} of 8 TB/s (median), {nbytes / (mn * 1e-6) / 8e12:.3f} (min)")


# ---- qK^T: in (B*nh, 1, D); kernel (B*nh, T/fpi, D); scale / zeros (B*nh, T/g, D)
fpi = 32 // bits
caches = []
for _ in range(NC):
    k = torch.randn((B, nh, T, D), device="cuda", dtype=torch.float16)
    code, scale, mn = new_pack.quantize_and_pack_k_tmajor(k, g, bits)         # hook layout (B, nh, D, T / fpi)
    del k
    caches.append((code.view(B * nh, D, -1).transpose(1, 2).contiguous(), scale.view(B * nh, D, -1).transpose(1, 2).contiguous(),
                   mn.view(B * nh, D, -1).transpose(1, 2).contiguous(), code, scale, mn))
q = torch.randn((B * nh, 1, D), device="cuda", dtype=torch.float16)
nbytes = B * nh * (D * T * bits // 8 + 2 * D * (T // g) * 2 + D * 2 + T * 2)
report(f"qK^T C2 {bits}-bit, reference kernel-input layout (gemv_outer_dim)", nbytes,
       lambda i: kivi_gemv.gemv_forward_cuda_outer_dim(q, caches[i][0], caches[i][1], caches[i][2], bits, g, nh, nh))
q4 = q.view(B, nh, 1, D)
report(f"qK^T C2 {bits}-bit, hook-state layout (kivi_gemv_k; no transposes)", nbytes,
       lambda i: matmul.cuda_bmm_fA_qB_outer(g, q4, caches[i][3], caches[i][4], caches[i][5], bits))
report("  the reference wrapper's three transposes for that call (matmul.py:205,213-214), torch", 2 * (nbytes - B * nh * (D * 2 + T * 2)),
       lambda i: (caches[i][3].view(B * nh, D, -1).transpose(1, 2).contiguous(), caches[i][4].view(B * nh, D, -1).transpose(1, 2).contiguous(),
                  caches[i][5].view(B * nh, D, -1).transpose(1, 2).contiguous()))
del caches
torch.cuda.empty_cache()
# ---- sV: in (B*nh, 1, Tv); kernel (B*nh, D/fpi, Tv); scale / zeros (B*nh, D/g, Tv)
Tv = 4064
caches = []
for _ in range(NC):
    v = torch.randn((B, nh, Tv, D), device="cuda", dtype=torch.float16)
    code, scale, mn = new_pack.triton_quantize_and_pack_along_last_dim(v, g, bits)   # (B, nh, Tv, D / fpi)
    del v
    caches.append((code.view(B * nh, Tv, -1).transpose(1, 2).contiguous(), scale.view(B * nh, Tv, -1).transpose(1, 2).contiguous(),
                   mn.view(B * nh, Tv, -1).transpose(1, 2).contiguous()))
a = torch.softmax(torch.randn((B * nh, 1, Tv), device="cuda"), -1).half()
nbytes = B * nh * (Tv * D * bits // 8 + 2 * Tv * (D // g) * 2 + Tv * 2 + D * 2)
report(f"sV C2 {bits}-bit, reference kernel-input layout (gemv_outer_dim)", nbytes,
       lambda i: kivi_gemv.gemv_forward_cuda_outer_dim(a, caches[i][0], caches[i][1], caches[i][2], bits, g, nh, nh))
