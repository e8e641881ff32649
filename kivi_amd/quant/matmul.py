"""Fused "fp16 vector x packed 2/4-bit matrix" batched GEMV over the KV cache.

Same names and signatures as the reference's quant/matmul.py.  Both entry
points run the same hand-written HIP kernels on the hook-state layout
(models/llama_kivi.py:454-455) -- the reference's three `.transpose(1, 2)
.contiguous()` copies per call (matmul.py:205, 213-214) do not exist here.
"""
from __future__ import annotations

import torch

from .. import _lib

__all__ = ["cuda_bmm_fA_qB_outer", "triton_bmm_fA_qB_outer", "gemv_k_paged", "bmm_variants", "bmm_fA_qB_outer_variant"]

_V_DIMS = {2: (64, 128, 256), 4: (32, 64, 128, 256)}

# Optional instrumentation (bench.py): called as hook("pre"|"post", kind, info) around each launch on the
# launch stream.  None in normal use.
launch_hook = None


def _prep(group_size, fA, qB, scales, zeros, bits):
    assert len(fA.shape) == 4 and len(qB.shape) == 4
    for t, n in ((fA, "fA"), (qB, "qB"), (scales, "scales"), (zeros, "zeros")):
        _lib.require_gpu(t, n)
    if fA.dtype != torch.float16 or scales.dtype != torch.float16 or zeros.dtype != torch.float16:
        raise TypeError("fA, scales and zeros must be float16 (the reference extension reads at::Half)")
    if qB.dtype != torch.int32:
        raise TypeError(f"qB must be int32, got {qB.dtype}")
    assert bits in [2, 4]                      # matmul.py:215
    B, nh, M, K = fA.shape
    nh_kv = qB.shape[1]
    assert nh % nh_kv == 0                     # matmul.py:216
    if M != 1:
        # the reference kernel ignores blockIdx.z and is only correct for M == 1 (gemv_cuda.cu:354-360)
        raise NotImplementedError("fused GEMV supports q_len == 1 (decode) only, like the reference kernel")
    fpi = 32 // bits
    N = qB.shape[-1] * fpi                     # matmul.py:204
    assert qB.shape[0] == B and qB.shape[2] == K
    assert scales.shape == zeros.shape == (B, nh_kv, K, N // group_size)
    if fA.stride(3) != 1:
        fA = fA.contiguous()
    if qB.stride(3) != 1:
        qB = qB.contiguous()
    if scales.stride(3) != 1:
        scales = scales.contiguous()
    if zeros.stride() != scales.stride():
        zeros = zeros.contiguous()
        scales = scales.contiguous()
    return fA, qB, scales, zeros, B, nh, nh_kv, K, N


def _run(group_size, fA, qB, scales, zeros, bits, variant=None, out=None):
    fA, qB, scales, zeros, B, nh, nh_kv, K, N = _prep(group_size, fA, qB, scales, zeros, bits)
    if out is None:
        out = torch.empty((B, nh, 1, N), dtype=torch.float16, device=fA.device)
    else:
        # in-place destination, e.g. the [..., :Tq] slice of a scores buffer (replaces the reference's torch.cat, :339)
        assert out.shape == (B, nh, 1, N) and out.dtype == torch.float16 and out.stride(3) == 1 and out.is_cuda
    lib = _lib.load()
    common = (_lib.ptr(fA), fA.stride(0), fA.stride(1),
              _lib.ptr(qB), qB.stride(0), qB.stride(1), qB.stride(2),
              _lib.ptr(scales), _lib.ptr(zeros), scales.stride(0), scales.stride(1), scales.stride(2),
              _lib.ptr(out), out.stride(0), out.stride(1))
    stream = _lib.stream_ptr(fA)
    if variant is None:
        # rows = reduction axis.  Long packed rows (qK^T: K = head_dim rows of T tokens) go to the
        # lanes-along-tokens kernel; short rows (sV: K = Tv rows of head_dim channels) to the
        # lanes-across-rows kernel.
        kind = "v" if (N in _V_DIMS[bits] and K >= N) else "k"
        vid = -1
    else:
        kind, vid = variant
    hook = launch_hook
    if hook is not None:
        info = dict(B=B, nh=nh, nh_kv=nh_kv, K=K, N=N, bits=bits, group_size=group_size)
        hook("pre", kind, info)
    if kind == "k":
        _lib.check(lib.kivi_gemv_k_variant(vid, *common, B, nh, nh_kv, K, N, group_size, bits, stream), "kivi_gemv_k")
    else:
        _lib.check(lib.kivi_gemv_v_variant(vid, *common, B, nh, nh_kv, K, N, group_size, bits, stream), "kivi_gemv_v")
    if hook is not None:
        hook("post", kind, info)
    return out


def cuda_bmm_fA_qB_outer(group_size: int, fA: torch.Tensor, qB: torch.Tensor, scales: torch.Tensor,
                         zeros: torch.Tensor, bits: int, out: torch.Tensor = None) -> torch.Tensor:
    """C = fA x dequant(qB) with groups along the OUTER (packed) dim (reference matmul.py:178-219).

    fA (B, nh, 1, K) fp16, qB (B, nh_kv, K, N // fpi) int32, scales / zeros (B, nh_kv, K, N // group_size) fp16
    -> (B, nh, 1, N) fp16.  C[b,h,0,n] = sum_k fA[b,h,0,k] * (scales[b,hk,k,n//g] * code[b,hk,k,n] + zeros[b,hk,k,n//g]),
    hk = h // (nh // nh_kv); fp32 arithmetic, one rounding to fp16.
    fA may be a last-dim-contiguous slice (llama_kivi.py:382); no copy is made.  `out` (optional, not in the
    reference signature) is a pre-allocated (B, nh, 1, N) destination view.
    """
    return _run(group_size, fA, qB, scales, zeros, bits, out=out)


def triton_bmm_fA_qB_outer(group_size: int, fA: torch.Tensor, qB: torch.Tensor, scales: torch.Tensor,
                           zeros: torch.Tensor, bits: int) -> torch.Tensor:
    """Alias of cuda_bmm_fA_qB_outer (reference matmul.py:112-175 is the same math in Triton; its
    group_size % 64 == 0 restriction does not apply here)."""
    return _run(group_size, fA, qB, scales, zeros, bits)


def gemv_k_paged(group_size: int, q: torch.Tensor, code_pages: torch.Tensor, scale_pages: torch.Tensor,
                 mn_pages: torch.Tensor, T: int, bits: int, out: torch.Tensor = None, variant: int = -1) -> torch.Tensor:
    """qK^T over PAGED per-channel K storage (kivi_amd.cache): code_pages (B, nh_kv, P, D, page_tokens // fpi) int32,
    scale_pages / mn_pages (B, nh_kv, P, D, page_tokens // group_size) fp16, first T tokens valid.
    Same arithmetic as cuda_bmm_fA_qB_outer on the equivalent (B, nh_kv, D, T // fpi) tensor."""
    for t, n in ((q, "q"), (code_pages, "code_pages"), (scale_pages, "scale_pages"), (mn_pages, "mn_pages")):
        _lib.require_gpu(t, n)
    assert bits in [2, 4]
    B, nh, M, D = q.shape
    if M != 1:
        raise NotImplementedError("fused GEMV supports q_len == 1 (decode) only, like the reference kernel")
    nh_kv = code_pages.shape[1]
    assert nh % nh_kv == 0
    fpi = 32 // bits
    page_tokens = code_pages.shape[4] * fpi
    assert code_pages.dim() == 5 and code_pages.shape[3] == D and code_pages.stride(4) == 1
    assert scale_pages.shape == mn_pages.shape and scale_pages.stride() == mn_pages.stride() and scale_pages.stride(4) == 1
    assert scale_pages.shape[4] * group_size == page_tokens and T <= code_pages.shape[2] * page_tokens
    if q.stride(3) != 1:
        q = q.contiguous()
    if out is None:
        out = torch.empty((B, nh, 1, T), dtype=torch.float16, device=q.device)
    else:
        assert out.shape == (B, nh, 1, T) and out.dtype == torch.float16 and out.stride(3) == 1 and out.is_cuda
    lib = _lib.load()
    hook = launch_hook
    if hook is not None:
        info = dict(B=B, nh=nh, nh_kv=nh_kv, K=D, N=T, bits=bits, group_size=group_size)
        hook("pre", "k", info)
    _lib.check(lib.kivi_gemv_k_paged(
        variant, page_tokens, code_pages.stride(2), scale_pages.stride(2),
        _lib.ptr(q), q.stride(0), q.stride(1),
        _lib.ptr(code_pages), code_pages.stride(0), code_pages.stride(1), code_pages.stride(3),
        _lib.ptr(scale_pages), _lib.ptr(mn_pages), scale_pages.stride(0), scale_pages.stride(1), scale_pages.stride(3),
        _lib.ptr(out), out.stride(0), out.stride(1), B, nh, nh_kv, D, T, group_size, bits, _lib.stream_ptr(q)),
        "kivi_gemv_k_paged")
    if hook is not None:
        hook("post", "k", info)
    return out


def bmm_variants(include_diagnostic: bool = False):
    """[(kind, id, name)] of every compiled kernel variant (bench / parity sweeps).  Variants named *_m3_* are
    memory-ceiling diagnostics that skip the unpack (wrong results) and are hidden unless asked for."""
    lib = _lib.load()
    out = [("k", i, lib.kivi_gemv_k_variant_name(i).decode()) for i in range(lib.kivi_gemv_k_num_variants())]
    out += [("v", i, lib.kivi_gemv_v_variant_name(i).decode()) for i in range(lib.kivi_gemv_v_num_variants())]
    return [v for v in out if include_diagnostic or "_m3_" not in v[2]]


def bmm_fA_qB_outer_variant(kind: str, vid: int, group_size, fA, qB, scales, zeros, bits):
    """Run one specific kernel variant (raises if it does not fit the problem)."""
    return _run(group_size, fA, qB, scales, zeros, bits, variant=(kind, vid))
