#!/usr/bin/env python3
"""The procedures of the reference's quant/test.py (:21-54 round trips, :173-202 fused GEMV vs matmul) on the
drop-in modules -- same import paths (`quant.new_pack`, `quant.matmul`), same calls, same printed quantities.

    python examples/quant_test.py        # needs an MI355X; there is no CPU fallback
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from quant.matmul import cuda_bmm_fA_qB_outer, triton_bmm_fA_qB_outer                                     # noqa: E402
from quant.new_pack import (quant_and_pack_kcache, triton_quantize_and_pack_along_last_dim,                # noqa: E402
                            unpack_and_dequant_kcache, unpack_and_dequant_vcache)


def test_vcache():            # quant/test.py:21-37
    torch.manual_seed(0)
    B, nh, T, hd = 55, 32, 433, 128
    v = torch.randn((B, nh, T, hd), device="cuda", dtype=torch.float16)
    group_size = 64
    for bits in [2, 4, 8]:
        code, scale, mn = triton_quantize_and_pack_along_last_dim(v, group_size, bits)
        dequant_v = unpack_and_dequant_vcache(code, scale.unsqueeze(-1), mn.unsqueeze(-1), group_size, bits)
        assert not dequant_v.isnan().any()
        gap = torch.nan_to_num((dequant_v - v) / v)
        print(f"bit {bits}, mean v rel arr: {torch.mean(torch.abs(gap))}")


def test_kcache():            # quant/test.py:40-54
    torch.manual_seed(0)
    BS, nh, T, D = 11, 32, 4096, 128
    k = torch.randn((BS, nh, T, D), device="cuda", dtype=torch.float16)
    group_size = 64
    for bits in [2, 4, 8]:
        code, scale, mn = triton_quantize_and_pack_along_last_dim(k.transpose(2, 3).contiguous(), group_size, bits)
        dequant_k = unpack_and_dequant_vcache(code, scale.unsqueeze(-1), mn.unsqueeze(-1), group_size, bits)
        assert not dequant_k.isnan().any()
        gap = torch.nan_to_num((dequant_k.transpose(2, 3) - k) / k)
        print(f"bit {bits}, k mean rel arr: {torch.mean(torch.abs(gap))}")


def test_4d_qmatmul():        # quant/test.py:173-202 (integer-valued inputs: the fused GEMV must be exact up to quantisation)
    torch.manual_seed(0)
    BS, nh, T, D = 16, 32, 1024, 128
    group_size = 64
    k = torch.randint(10, (BS, nh, T, D), device="cuda").to(torch.float16)
    query_state = torch.randint(5, (BS, nh, 1, D), device="cuda").to(torch.float16)
    for bits in [4, 2]:       # the reference's CUDA kernel is 2 / 4 bit (matmul.py:215)
        code, scale, mn = quant_and_pack_kcache(k, group_size, bits)
        dequant_k = unpack_and_dequant_kcache(code, scale, mn, group_size, bits)
        code = code.transpose(2, 3)
        scale = scale.view(BS, nh, -1, D).transpose(2, 3)
        mn = mn.view(BS, nh, -1, D).transpose(2, 3)
        for name, fn in (("cuda_bmm_fA_qB_outer", cuda_bmm_fA_qB_outer), ("triton_bmm_fA_qB_outer", triton_bmm_fA_qB_outer)):
            our_out = fn(group_size, query_state, code.contiguous(), scale.contiguous(), mn.contiguous(), bits)
            ref_out = torch.matmul(query_state, k.transpose(2, 3))
            fake = torch.matmul(query_state, dequant_k.transpose(2, 3))
            assert not our_out.isnan().any() and not ref_out.isnan().any()
            err = torch.mean(torch.abs(torch.nan_to_num((our_out - ref_out) / ref_out))).item()
            err_fake = torch.mean(torch.abs(torch.nan_to_num((our_out - fake) / fake))).item()
            print(f"{name} bits {bits}, err vs fp16 matmul: {err:.3e}, vs dequantised matmul: {err_fake:.3e}")


if __name__ == "__main__":
    test_vcache()
    test_kcache()
    test_4d_qmatmul()
