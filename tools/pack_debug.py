#!/usr/bin/env python3
"""Debug aid: compare the packed-16-bit 2-bit last-dim pack kernel with the scalar one (KIVI_PACK_NO_PK16=1) on the
hard-value tensor of tests/test_pack_gpu.py::test_lastdim_hard_values and print every group whose codes differ."""
import os, sys, subprocess
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kivi_amd.quant import new_pack

def hard():
    g = torch.Generator().manual_seed(5)
    x = (torch.randn((1, 4, 64, 128), generator=g) * torch.exp(4 * torch.randn((1, 4, 64, 128), generator=g))).half()
    x[0, 0, 0, :32] = 0.0
    x[0, 0, 1, :32] = 3.25
    x[0, 0, 2, :32] = torch.tensor([0.0, 1.0, 2.0, 3.0] * 8)
    x[0, 0, 3, :32] = torch.tensor([0.0, 0.5, 1.5, 2.5, 3.0, 1.0, 2.0, 0.25] * 4)
    x[0, 0, 4, :32] = torch.tensor([6e-8, 1.2e-7, 0.0, 5.9e-8] * 8)
    x[0, 0, 5, :32] = torch.tensor([65504.0, -65504.0] * 16)
    x[0, 0, 6, :32] = torch.tensor([-0.0, 0.0] * 16)
    return x

x = hard()
if len(sys.argv) > 1:   # child: dump the codes of this mode
    code, scale, mn = new_pack.triton_quantize_and_pack_along_last_dim(x.cuda(), 32, 2)
    np.save(sys.argv[1], code.cpu().numpy())
    sys.exit(0)
subprocess.check_call([sys.executable, __file__, "/tmp/_pk_new.npy"])
subprocess.check_call([sys.executable, __file__, "/tmp/_pk_old.npy"], env=dict(os.environ, KIVI_PACK_NO_PK16="1"))
new, old = np.load("/tmp/_pk_new.npy").reshape(-1, 2), np.load("/tmp/_pk_old.npy").reshape(-1, 2)
xs = x.reshape(-1, 32)
bad = np.nonzero((new != old).any(axis=1))[0]
print(f"{len(bad)} of {len(new)} groups differ")
for gi in bad[:6]:
    xv = xs[gi]
    print("group", gi, "x bits", [hex(int(b)) for b in xv.view(torch.int16).numpy().astype(np.uint16)])
    cn = [(int(new[gi, i // 16]) >> (2 * (i % 16))) & 3 for i in range(32)]
    co = [(int(old[gi, i // 16]) >> (2 * (i % 16))) & 3 for i in range(32)]
    print("   new", cn)
    print("   old", co)
