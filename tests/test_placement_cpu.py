"""CPU: the arithmetic of the big-scale placement (kivi_amd/csrc/kivi_mf_dev.h: mf_sp / mf_ksh / mf_big_d) restated and checked over every case --
the three properties the kernels rely on: no probability is rounded by the part of the shift that goes into p'', the operand p'' * scale stays
finite for every finite fp16 scale, and the two parts always add up to the shift the unit needs."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _const(name):
    src = open(os.path.join(ROOT, "kivi_amd", "csrc", "kivi_mfma_layout.h")).read()
    return int(re.search(rf"#define {name} (\d+)", src).group(1))


def test_split_of_the_big_scale_shift():
    V = _const("KIVI_MF_BIG_SHIFT_V")
    assert V == 7
    src = open(os.path.join(ROOT, "kivi_amd", "csrc", "kivi_mf_dev.h")).read()
    assert "return (e + 4 < KIVI_MF_BIG_SHIFT_V) ? e + 4 : KIVI_MF_BIG_SHIFT_V;" in src        # mf_big_d, restated below
    assert "return rsh < 0 ? e - mf_big_d(e) : e + rsh;" in src                                  # mf_sp
    for e in range(0, 15):                                           # e = clamp(floor(log2 sum), 0, 14)
        d = min(V, e + 4)
        ksh = V - d
        sp = e - d
        assert d + ksh == V and 0 <= ksh <= 3 and -4 <= sp <= 7
        # p'' = p * 2^(sp + a), a = 4 (registers 0, 1) or 6: a non-negative power of two -> exact for every fp16 p, subnormals included
        assert sp + 4 >= 0
        # the largest probability of the row is <= 2^-e (p <= 1 / sum): p'' <= 2^(6 - d); scales enter 2^-ksh times their value
        worst = 2.0 ** (sp + 6 - e) * 65504.0 * 2.0 ** -ksh
        assert worst <= 32752.0, (e, worst)                          # finite in fp16 (65504) with a factor of two to spare
        # kivi_gqa_output places by the largest |p| instead of the sum: 2^e * max|p| < 2, i.e. twice the bound above -- still finite
        assert 2 * worst <= 65504.0
        if e >= 3:
            assert ksh == 0                                          # a row whose sum is >= 8 leaves the scales alone
