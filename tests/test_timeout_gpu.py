"""GPU: a sliced one-launch decode step whose partner block never arrives (fault injection of the -DKIVI_TUNING build:
KIVI_MF_FAULT_DROP_ARRIVAL=1 keeps slice 0 of unit 0 from announcing itself) must not fail silently (advisor r5):
  * the waiting blocks give up after ~1 s, unit 0's output of that step is NaN, every other unit is untouched;
  * the step is REPORTED: kivi_device_error() / the next decode call returns KIVI_ETIMEOUT (KiviTimeout) once, naming the unit;
  * the launch's last blocks put every arrival counter and the ticket back to zero: the call after that, on the same workspace,
    runs normally and agrees with an untouched clone.
Runs in a subprocess: the tuning library is selected at import time (KIVI_TUNING=1 KIVI_HIP_LIB=...; __graft_entry__.build() builds it)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TUNING_LIB = os.path.join(ROOT, "kivi_amd", "_variants", "libkivi_tuning.so")

SCRIPT = r'''
import sys, torch
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
from helpers import make_kv
from kivi_amd import _lib
from kivi_amd.attention import KiviConfig, kivi_attention_decode, make_layer_cache
R_, nh, nh_kv = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
B, T0, D = 2, 1500, 128
cfg = KiviConfig(2, 2, 32, R_)
k0, v0 = make_kv(1, B, nh_kv, T0, D).cuda(), make_kv(2, B, nh_kv, T0, D).cuda()
a = make_layer_cache(cfg, B, nh_kv, D, T0 + 64, "cuda", num_heads=nh)
a.prefill(k0, v0)
ref = a.clone()
lib = _lib.load()
assert lib.kivi_device_error() == 0
q, kn, vn = make_kv(3, B, nh, 1, D).cuda(), make_kv(4, B, nh_kv, 1, D).cuda(), make_kv(5, B, nh_kv, 1, D).cuda()
ref.flags = _lib.GQA_FORCE_SPLIT
want = kivi_attention_decode(q, kn, vn, ref)                    # two launches: no slices, no fault
a.flags = _lib.gqa_slices(2)
got = kivi_attention_decode(q, kn, vn, a)                       # unit 0's slice 0 never arrives -> ~1 s, then NaN
torch.cuda.synchronize()
r = nh // nh_kv
assert torch.isnan(got[0, :r]).all(), "unit 0 must be poisoned"
assert torch.equal(got[0, r:], want[0, r:]) or (got[0, r:].float() - want[0, r:].float()).abs().max() < 2e-2, "other units untouched"
assert torch.isfinite(got[1]).all()
ws = a._native[4][2].view(torch.int32)
assert int(ws[16375]) == 4, "device error word of the workspace"
assert int(ws[:16375].abs().sum()) == 0 and int(ws[16376:16384].abs().sum()) == 0, "arrival counters / the eight tickets back at zero"
q2, kn2, vn2 = make_kv(6, B, nh, 1, D).cuda(), make_kv(7, B, nh_kv, 1, D).cuda(), make_kv(8, B, nh_kv, 1, D).cuda()
try:
    kivi_attention_decode(q2, kn2, vn2, a)
    raise SystemExit("the step after a timeout must report it")
except _lib.KiviTimeout as e:
    assert e.rc == -4 and "unit 0" in str(e), str(e)
assert lib.kivi_device_error() == 0                             # reported once, cleared
a.flags = _lib.GQA_FORCE_SPLIT                                  # same workspace, a form the fault does not touch
got2 = kivi_attention_decode(q2, kn2, vn2, a)
want2 = kivi_attention_decode(q2, kn2, vn2, ref)
torch.cuda.synchronize()
assert torch.equal(got2, want2), "the call after the report runs normally"
print("TIMEOUT PATH OK")
'''


@pytest.mark.parametrize("R,nh,nh_kv", [(32, 16, 4), (128, 16, 2), (32, 4, 4)])
def test_missing_partner_is_reported_and_the_workspace_recovers(R, nh, nh_kv):
    if not os.path.exists(TUNING_LIB):
        pytest.skip("no tuning build (tools/build_variant.sh tuning -DKIVI_TUNING; __graft_entry__.build() makes it)")
    env = dict(os.environ, KIVI_TUNING="1", KIVI_HIP_LIB=TUNING_LIB, KIVI_MF_FAULT_DROP_ARRIVAL="1")
    p = subprocess.run([sys.executable, "-c", SCRIPT, ROOT, str(R), str(nh), str(nh_kv)], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0 and "TIMEOUT PATH OK" in p.stdout, (p.stdout[-2000:], p.stderr[-3000:])
