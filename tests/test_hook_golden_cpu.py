"""The hook restatement (oracle/hook_ref.py) against fixtures produced by the REFERENCE's own attention classes
(oracle/pin_hook.py executes LlamaAttention_KIVI / LlamaFlashAttention_KIVI from models/llama_kivi.py on CPU)."""
import glob
import os

import numpy as np
import pytest
import torch

from helpers import gemv_close, same_bits

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "hook_*.npz")))
NAMES = ["K_code_T", "K_full", "K_scale_T", "K_mn_T", "V_code", "V_full", "V_scale", "V_mn"]


def load_case(path):
    z = np.load(path)
    B, nh, nh_kv, D, bits, g, R, T0, steps, masked = (int(x) for x in z["cfg"])
    masks = [None] * steps
    if masked:
        for s in range(steps):
            m = torch.zeros((B, 1, 1, T0 + s + 1), dtype=torch.float16)
            m[0, :, :, : int(z["mask_prefix"][s])] = torch.finfo(torch.float16).min
            masks[s] = m
    final = [torch.from_numpy(z["final_" + n]) if ("final_" + n) in z.files else None for n in NAMES]
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a))  # noqa: E731
    return dict(B=B, nh=nh, nh_kv=nh_kv, D=D, bits=bits, g=g, R=R, T0=T0, steps=steps, masks=masks, k0=t(z["k0"]),
                v0=t(z["v0"]), q=t(z["q"]), k=t(z["k"]), v=t(z["v"]), out=t(z["out"]), final=final,
                final_len=int(z["final_len"][0]))


def test_fixtures_present():
    assert len(GOLDEN) >= 4, "run oracle/pin_hook.py in the build container"


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[5:-4] for p in GOLDEN])
def test_hook_restatement_matches_reference_classes(oracle, path):
    from oracle import hook_ref as H
    c = load_case(path)
    past = H.prefill_cache(c["k0"], c["v0"], c["bits"], c["bits"], c["g"], c["R"])
    for s in range(c["steps"]):
        out, past = H.decode_step(c["q"][s], c["k"][s], c["v"][s], past, c["bits"], c["bits"], c["g"], c["R"],
                                  attention_mask=c["masks"][s])
        ok, ratio = gemv_close(out, c["out"][s], rtol=2e-3, ulps=1)
        assert ok, (s, ratio)
    for n, a, b in zip(NAMES, past[:8], c["final"]):
        if b is None:
            assert a is None or a.numel() == 0, n
        else:
            assert a is not None and same_bits(a, b), n
    assert past[8] == c["final_len"]
