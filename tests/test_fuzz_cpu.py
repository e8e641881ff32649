"""CPU: tools/fuzz_decode.py stays importable without a GPU and its case generator is a pure function of (seed, index) -- the lines of
profiles/r06_fuzz_seed*.log name their cases by that pair (`--only SEED:INDEX` re-runs one)."""
import ast
import importlib.util
import os
import random
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _tool():
    spec = importlib.util.spec_from_file_location("fuzz_decode", os.path.join(ROOT, "tools", "fuzz_decode.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_cases_of_the_committed_logs_are_reproducible():
    fz = _tool()
    seen = 0
    for seed in (1, 2, 3):
        with open(os.path.join(ROOT, "profiles", f"r06_fuzz_seed{seed}.log")) as f:
            for line in f:
                m = re.match(r"^(\d+):(\d+) (\{.*?\}) ", line)
                if not m or int(m.group(2)) % 25:
                    continue
                cfg = ast.literal_eval(m.group(3))
                assert fz.draw(random.Random(int(m.group(1)) * 1_000_003 + int(m.group(2)))) == cfg, line[:80]
                seen += 1
    assert seen >= 20


def test_generator_covers_both_layouts_and_the_plan_boundaries():
    fz = _tool()
    cases = [fz.draw(random.Random(7 * 1_000_003 + i)) for i in range(400)]
    assert {c["dist"] for c in cases} == {"randn", "outlier", "small", "big", "mixed"}
    assert {c["nh"] // c["nh_kv"] for c in cases} == {1, 2, 4, 8} and {c["g"] for c in cases} == {32, 64} and {c["D"] for c in cases} == {64, 128}
    assert any(c["k_bits"] != c["v_bits"] for c in cases) and any(c["flags"] for c in cases) and any(c["masked"] for c in cases)
    assert any(8192 - 140 <= c["T0"] <= 8192 + 140 for c in cases) and any(c["T0"] <= 2 * c["R"] + 2 for c in cases)
    assert all(c["R"] % c["g"] == 0 and c["R"] <= 128 and c["B"] * c["nh_kv"] * (c["T0"] + 200) <= 3_000_000 + 48 * 32 * 200 for c in cases)
