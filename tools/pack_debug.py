#!/usr/bin/env python3
"""Debug aid for the 2-bit last-dim pack kernels: run the packed-16-bit kernel and the scalar one (KIVI_PACK_NO_PK16=1)
on the tensors of tests/test_pack_gpu.py (hard values, every-exponent grid) and print every group whose scale / mn /
codes differ from the CPU oracle."""
import os, sys, subprocess
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
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

def grid(g=32, rows=37):
    gen = torch.Generator().manual_seed(17)
    ngrp = rows * 128 // g
    e = torch.randint(-24, 16, (ngrp, 1), generator=gen).float()
    base = torch.randint(0, 2048, (ngrp, 1), generator=gen).float()
    x = ((torch.randint(-24, 25, (ngrp, g), generator=gen).float() + base) * torch.exp2(e - 5)).half()
    return x.reshape(1, 1, rows, 128)

G = int(os.environ.get("G", "32"))
x = hard() if os.environ.get("CASE", "grid") == "hard" else grid(G)
if len(sys.argv) > 1:   # child: dump the outputs of this mode
    code, scale, mn = new_pack.triton_quantize_and_pack_along_last_dim(x.cuda(), G, 2)
    np.savez(sys.argv[1], code=code.cpu().numpy(), scale=scale.cpu().view(torch.int16).numpy(), mn=mn.cpu().view(torch.int16).numpy())
    sys.exit(0)
subprocess.check_call([sys.executable, __file__, "/tmp/_pk_new.npz"])
subprocess.check_call([sys.executable, __file__, "/tmp/_pk_old.npz"], env=dict(os.environ, KIVI_PACK_NO_PK16="1"))
new, old = np.load("/tmp/_pk_new.npz"), np.load("/tmp/_pk_old.npz")
wpg = G // 16
cn, co = new["code"].reshape(-1, wpg), old["code"].reshape(-1, wpg)
sn, so, mn_, mo = new["scale"].reshape(-1), old["scale"].reshape(-1), new["mn"].reshape(-1), old["mn"].reshape(-1)
xs = x.reshape(-1, G)
bad = np.nonzero((cn != co).any(axis=1) | (sn != so) | (mn_ != mo))[0]
print(f"{len(bad)} of {len(cn)} groups differ")
for gi in bad[:6]:
    xv = xs[gi]
    print("group", gi, "scale new/old", hex(int(sn[gi]) & 0xFFFF), hex(int(so[gi]) & 0xFFFF), "mn", hex(int(mn_[gi]) & 0xFFFF), hex(int(mo[gi]) & 0xFFFF))
    print("   x bits", [hex(int(b)) for b in xv.view(torch.int16).numpy().astype(np.uint16)])
    print("   new", [(int(cn[gi, i // 16]) >> (2 * (i % 16))) & 3 for i in range(G)])
    print("   old", [(int(co[gi, i // 16]) >> (2 * (i % 16))) & 3 for i in range(G)])
