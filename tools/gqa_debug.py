import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kivi_amd.quant import matmul, mfma, new_pack
torch.manual_seed(0)
B, nh, nh_kv, T = 1, 4, 1, 64
k = torch.randn(B, nh_kv, T, 128).half().cuda()
q = torch.randn(B, nh, 1, 128).half().cuda()
store = mfma.alloc_store(B, nh_kv, 1, "cuda")
mfma.kt_pack(k, store, 0)
out = torch.zeros((B, nh, 1, T), dtype=torch.float16, device="cuda")
mfma.gqa_scores(q, store, T, out)
code, scale, mn = new_pack.quantize_and_pack_k_tmajor(k, 32, 2)
ref = matmul.cuda_bmm_fA_qB_outer(32, q, code, scale, mn, 2)
o, r = out[0, :, 0].float().cpu(), ref[0, :, 0].float().cpu()
torch.set_printoptions(precision=3, linewidth=220, sci_mode=False)
print("out head0", o[0, :40]); print("ref head0", r[0, :40])
# which ref (head, token) does each out (head, token) match best?
for h in range(nh):
    d = (o[h][:, None, None] - r[None]).abs()          # (T, nh, T)
    best = d.view(T, -1).argmin(1)
    print("head", h, "best match (head,token) per token:", [(int(x) // T, int(x) % T) for x in best[:34]])
# zero-point only and scale-only decomposition
deq = new_pack.unpack_and_dequant_vcache(code, scale.unsqueeze(-1), mn.unsqueeze(-1), 32, 2)   # (B,nh_kv,D,T)
zp = torch.einsum("hd,dg->hg", q[0, :, 0].float(), mn[0, 0].float())
print("zero-point term per group head0:", zp[0])
print("out - ref head0:", (o - r)[0, :40])
print("ratio out/ref head0:", (o / r)[0, :20])
