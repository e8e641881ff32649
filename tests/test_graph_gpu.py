"""GPU: decode steps with DEVICE-RESIDENT lengths (kivi_mf_decode_layer_dyn, include/kivi_hip.h: kivi_mf_step) and their replay
from a hipGraph (kivi_amd/graph.py) against the eager layer step (kivi_mf_decode_layer): same kernels, same arithmetic -> outputs
and cache tuples BIT-identical, step after step, through K flushes, V flushes, the window ring wrapping, changes of the geometry
class (a 512-token boundary of either store) and cache growth -- in every form: two launches, a block per row, rows cut into slices
(whose geometry the blocks derive from the device-resident lengths).  The eager step itself is pinned against the reference logic in
tests/test_mfma_gpu.py / tests/test_hook_gpu.py.  (Eager and replayed steps follow the same launch plan -- that of the step's geometry
class -- everywhere, incl. Tq just under 8192 (nh == nh_kv) and 9216 (nh / nh_kv = 4): the last parameter sets.)"""
import pytest
import torch

from helpers import make_kv, same_bits

pytestmark = pytest.mark.gpu


def _tuples_equal(a, b):
    ta, tb = a.as_tuple(), b.as_tuple()
    for x, y in zip(ta[:8], tb[:8]):
        assert (x is None and y is None) or (x is not None and y is not None and same_bits(x, y))
    assert ta[8] == tb[8]


@pytest.mark.parametrize("B,nh,nh_kv,T0,R,form,masked,bits", [(2, 4, 4, 70, 32, "row", False, 2), (2, 8, 2, 1100, 128, "row", True, 2),
                                                                (2, 4, 4, 600, 64, "split", True, 2), (2, 8, 2, 500, 32, "split", False, 2),
                                                                (1, 16, 2, 480, 32, "split", False, 2), (2, 16, 2, 300, 32, "row", False, 2),
                                                                (8, 32, 32, 4080, 32, "auto", False, 2),
                                                                (2, 8, 2, 460, 32, "row", True, 4), (2, 8, 2, 500, 32, "split", False, 4),
                                                                (2, 8, 2, 1100, 128, "slices2", True, 2), (2, 16, 2, 1500, 32, "slices3", False, 2),
                                                                # the bands of advisor r4 / r5: the class bound (ceil(Tq / 512) * 512 + R) exceeds
                                                                # 8192 / 9216 keys while the step's own row does not -- a block per row in both
                                                                # (caps = whole super-blocks + a full residual), and the plan's own choice
                                                                (2, 4, 4, 8000, 32, "row", False, 2), (2, 8, 2, 9000, 128, "row", True, 2),
                                                                (2, 4, 4, 8100, 32, "auto", True, 2), (2, 8, 2, 9100, 128, "auto", False, 2),
                                                                (2, 4, 4, 8130, 32, "auto", False, 2)])
def test_dyn_steps_bit_identical_to_eager(B, nh, nh_kv, T0, R, form, masked, bits):
    from kivi_amd import _lib
    from kivi_amd.attention import KiviConfig, kivi_attention_decode, make_layer_cache
    from kivi_amd.graph import MfStepDriver
    D = 128
    cfg = KiviConfig(bits, bits, 32, R)
    k0, v0 = make_kv(1, B, nh_kv, T0, D, "outlier").cuda(), make_kv(2, B, nh_kv, T0, D).cuda()
    a = make_layer_cache(cfg, B, nh_kv, D, T0 + 8, "cuda", num_heads=nh)          # small capacity: both must grow on the way
    a.flags = _lib.gqa_slices(int(form[6:])) if form.startswith("slices") else {"row": _lib.GQA_FORCE_ROW, "split": _lib.GQA_FORCE_SPLIT, "auto": 0}[form]
    a.prefill(k0, v0)
    b = a.clone()
    drv = MfStepDriver([b])
    steps = R + 41
    pitch = ((T0 + steps + 8 + 7) // 8) * 8
    mask_buf = torch.zeros((B, 1, 1, pitch), dtype=torch.float16, device="cuda")
    out_b = torch.empty((B, nh, 1, D), dtype=torch.float16, device="cuda")
    keys = set()
    for s in range(steps):
        q, kn, vn = make_kv(100 + s, B, nh, 1, D).cuda(), make_kv(200 + s, B, nh_kv, 1, D, "outlier").cuda(), make_kv(300 + s, B, nh_kv, 1, D).cuda()
        n = T0 + s + 1
        mask = None
        if masked:
            mask_buf.zero_()
            mask_buf[0, ..., : min(9, n - 1)] = torch.finfo(torch.float16).min
            mask = mask_buf[..., :n].contiguous()
        oa = kivi_attention_decode(q, kn, vn, a, attention_mask=mask)
        drv.prepare()
        keys.add(drv.key())
        drv.enqueue(0, q, kn, vn, out_b, mask_buf if masked else None)
        drv.finish()
        assert torch.equal(oa, out_b), s
        if s % 16 == 0 or s == steps - 1:
            _tuples_equal(a, b)
    sb = lambda t: (t + 511) // 512                          # noqa: E731
    kv0, kv1 = T0, T0 + steps - 1
    crossings = (sb((kv1 // R) * R) - sb((kv0 // R) * R)) + (sb(max(kv1 - R, 0)) - sb(max(kv0 - R, 0)))
    assert len(keys) >= 1 + crossings, (keys, crossings)     # a store gaining a super-block starts a new geometry class


@pytest.mark.parametrize("B,nh,nh_kv,T0,R,bits", [(2, 4, 4, 460, 32, 2), (2, 8, 2, 900, 128, 2), (4, 32, 32, 4080, 32, 2), (1, 32, 32, 8100, 32, 2),
                                                  (2, 8, 2, 900, 128, 4)])
def test_graph_replay_bit_identical_to_eager(B, nh, nh_kv, T0, R, bits):
    """Two layers replayed from one hipGraph (kivi_amd.graph.GraphedDecode) against the eager steps on cloned caches."""
    from kivi_amd.attention import KiviConfig, kivi_attention_decode, make_layer_cache
    from kivi_amd.graph import GraphedDecode, MfStepDriver
    D, L = 128, 2
    cfg = KiviConfig(bits, bits, 32, R)
    eager, graphed = [], []
    for layer in range(L):
        k0, v0 = make_kv(10 + layer, B, nh_kv, T0, D).cuda(), make_kv(20 + layer, B, nh_kv, T0, D).cuda()
        c = make_layer_cache(cfg, B, nh_kv, D, T0 + 700, "cuda", num_heads=nh)
        c.prefill(k0, v0)
        eager.append(c)
        graphed.append(c.clone())
    drv = MfStepDriver(graphed)
    qs = [torch.zeros((B, nh, 1, D), dtype=torch.float16, device="cuda") for _ in range(L)]
    ks = [torch.zeros((B, nh_kv, 1, D), dtype=torch.float16, device="cuda") for _ in range(L)]
    vs = [torch.zeros((B, nh_kv, 1, D), dtype=torch.float16, device="cuda") for _ in range(L)]
    outs = [torch.zeros((B, nh, 1, D), dtype=torch.float16, device="cuda") for _ in range(L)]

    def step_fn():
        for i in range(L):
            drv.enqueue(i, qs[i], ks[i], vs[i], outs[i])

    gd = GraphedDecode(drv, step_fn)
    steps = 100
    for s in range(steps):
        for i in range(L):
            qs[i].copy_(make_kv(1000 + 7 * s + i, B, nh, 1, D).cuda())
            ks[i].copy_(make_kv(2000 + 7 * s + i, B, nh_kv, 1, D).cuda())
            vs[i].copy_(make_kv(3000 + 7 * s + i, B, nh_kv, 1, D).cuda())
        ref = [kivi_attention_decode(qs[i], ks[i], vs[i], eager[i]) for i in range(L)]
        gd.step()
        torch.cuda.synchronize()
        for i in range(L):
            assert torch.equal(ref[i], outs[i]), (s, i)
    for i in range(L):
        _tuples_equal(eager[i], graphed[i])
    assert gd.replays >= steps - 2 * gd.captures - 4 and gd.captures >= 1, (gd.eager, gd.captures, gd.replays)
    print("eager / captures / replays:", gd.eager, gd.captures, gd.replays)
