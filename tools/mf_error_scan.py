"""Error of the matrix-pipe sV (kivi_gqa_output) and qK^T (kivi_gqa_scores) against the fp32 VALU kernels of the reference layout as a function
of the row length, in units of the 1e-3 GEMV bar (0.49 = the fp16 rounding of the result): the inputs of tests/test_mfma_gpu.py::
test_gqa_output_vs_oracle ('outlier' values + peaked softmax rows, uniform rows) for nh / nh_kv = 4, 1, 8 and T = 2k .. 32k.  Used to
compare cache-layout / operand-format variants on one box (KIVI_TUNING=1 KIVI_HIP_LIB=<other build>): profiles/r05_inplace_fields.log.
Run from the repo root on a GPU box:  python tools/mf_error_scan.py"""
import sys, torch
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from helpers import gemv_close, make_kv
from kivi_amd.quant import matmul, mfma, new_pack
from test_mfma_gpu import _probs

for (B, nh, nh_kv) in [(1, 32, 8), (1, 4, 4), (1, 8, 1)]:
    for kind in ["softmax", "uniform"]:
        for T in [2048, 4096, 8192, 16384, 32768]:
            v = make_kv(21, B, nh_kv, T, 128, "outlier" if kind == "softmax" else "randn").cuda()
            store = mfma.alloc_store(B, nh_kv, (T + 511) // 512, "cuda")
            mfma.vt_pack(v, store)
            pitch = (T + 7) // 8 * 8 + 8
            probs = torch.zeros((B, nh, 1, pitch), dtype=torch.float16, device="cuda")
            probs[..., :T] = _probs(kind, B, nh, T, 5).cuda()
            out = mfma.gqa_output(probs, store, T)
            code, scale, mn = new_pack.triton_quantize_and_pack_along_last_dim(v, 32, 2)
            ref = matmul.cuda_bmm_fA_qB_outer(32, probs[..., :T], code, scale, mn, 2)
            ok, ratio = gemv_close(out, ref.cpu(), rtol=1e-3)
            # scores
            k = make_kv(3, B, nh_kv, T, 128, "outlier").cuda()
            q = make_kv(4, B, nh, 1, 128).cuda()
            ks = mfma.alloc_store(B, nh_kv, (T + 511) // 512, "cuda")
            mfma.kt_pack(k, ks, 0)
            kc, ksc, kmn = new_pack.quantize_and_pack_k_tmajor(k, 32, 2)
            so = torch.zeros((B, nh, 1, T + 8), dtype=torch.float16, device="cuda")
            mfma.gqa_scores(q, ks, T, so)
            sref = matmul.cuda_bmm_fA_qB_outer(32, q, kc, ksc, kmn, 2)
            ok2, ratio2 = gemv_close(so[..., :T], sref.cpu(), rtol=1e-3)
            print(f"R={nh // nh_kv} {kind:8s} T={T:6d}  sV err/1e-3 bar = {ratio:.3f}   qK err/1e-3 bar = {ratio2:.3f}", flush=True)
