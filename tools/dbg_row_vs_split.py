import sys, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from kivi_amd import _lib
from kivi_amd.attention import KiviConfig, kivi_attention_decode, make_layer_cache
def run(B, nh, nh_kv, T0, R, steps=3):
    D, g = 128, 32
    cfg = KiviConfig(2, 2, g, R)
    gen = torch.Generator().manual_seed(1)
    k0 = torch.randn((B, nh_kv, T0, D), generator=gen).half(); v0 = torch.randn((B, nh_kv, T0, D), generator=gen).half()
    a = make_layer_cache(cfg, B, nh_kv, D, T0 + 2 * R + 8, "cuda", num_heads=nh)
    a.prefill(k0.cuda(), v0.cuda())
    b_ = a.clone()
    a.flags, b_.flags = _lib.GQA_FORCE_ROW, _lib.GQA_FORCE_SPLIT
    import os
    os.environ["KIVI_NO_MFMA_LAYOUT"] = "1"
    c_ = make_layer_cache(cfg, B, nh_kv, D, T0 + 2 * R + 8, "cuda", num_heads=nh)
    del os.environ["KIVI_NO_MFMA_LAYOUT"]
    c_.prefill(k0.cuda(), v0.cuda())
    print(type(a).__name__, type(c_).__name__)
    for s in range(steps):
        q = torch.randn((B, nh, 1, D), generator=gen).half().cuda(); kn = torch.randn((B, nh_kv, 1, D), generator=gen).half().cuda(); vn = torch.randn((B, nh_kv, 1, D), generator=gen).half().cuda()
        oa = kivi_attention_decode(q, kn, vn, a); ob = kivi_attention_decode(q, kn, vn, b_); oc = kivi_attention_decode(q, kn, vn, c_)
        d = (oa.float() - ob.float()).abs()
        print((B, nh, nh_kv, T0, R), "step", s, "max diff", d.max().item(), "ref max", ob.float().abs().max().item(), "nan", torch.isnan(oa).any().item(), "| row vs hook", (oa.float() - oc.float()).abs().max().item(), "split vs hook", (ob.float() - oc.float()).abs().max().item())
        if d.max() > 0.05:
            idx = (d > 0.05).nonzero()
            print(" bad count", idx.shape[0], "of", d.numel(), "first", idx[:6].tolist())
            i0 = idx[0].tolist()
            print(" oa", oa[i0[0], i0[1], 0, :8].tolist()); print(" ob", ob[i0[0], i0[1], 0, :8].tolist()); print(" oc", oc[i0[0], i0[1], 0, :8].tolist())
for cfg in [(2, 4, 4, 5, 32), (2, 4, 4, 100, 32), (2, 4, 4, 1000, 32), (40, 32, 32, 1000, 32), (2, 8, 2, 5, 32), (2, 8, 2, 1000, 32)]:
    run(*cfg)
