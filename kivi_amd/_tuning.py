"""The ONE place where product Python may look at tuning / A-B environment knobs -- and only when KIVI_TUNING=1.

The C library reads no environment variable at all (tuning variants exist only in -DKIVI_TUNING builds); the Python side
keeps a handful of A/B switches for the tools under tools/ (cache layout, decode fusion level).  They are inert unless the
process is started with KIVI_TUNING=1, so a production process behaves the same whatever else its environment holds;
tests/test_abi_cpu.py asserts that no other product module reads os.environ (besides _lib.py's KIVI_HIP_LIB, the path of
an alternative build, honoured under the same switch).
"""
from __future__ import annotations

import os

ENABLED = os.environ.get("KIVI_TUNING") == "1"


def knob(name: str, default=None):
    """Value of the environment knob `name` in a tuning session, `default` otherwise."""
    return os.environ.get(name, default) if ENABLED else default


def flag(name: str) -> bool:
    """True when the knob is set to anything but "" / "0" in a tuning session."""
    return knob(name) not in (None, "", "0")
