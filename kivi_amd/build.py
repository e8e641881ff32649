"""Build libkivi_hip.so (hand-written HIP kernels + C ABI) for gfx950 with hipcc.

In-tree build: the .so lands next to this file so it travels with the repo
snapshot to the GPU box.  No torch headers are involved; the library is a
plain C-ABI shared object (include/kivi_hip.h).

    python -m kivi_amd.build [--force]
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libkivi_hip.so")
SOURCES = ["kivi_abi.hip", "kivi_pack.hip", "kivi_gemv_k.hip", "kivi_gemv_v.hip", "kivi_gemv_compat.hip",
           "kivi_softmax.hip", "kivi_layer.hip", "kivi_gqa.hip", "kivi_mf.hip"]
import glob

HEADERS = sorted(glob.glob(os.path.join(CSRC, "*.h"))) + [os.path.join(os.path.dirname(HERE), "include", "kivi_hip.h")]

FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
    # bit-exact quantisation needs IEEE fp32 division; never fast-math
    "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-fast-math",
    "-Wall", "-Wno-unused-function",
]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm)")


TUNING_LIB = os.path.join(HERE, "_variants", "libkivi_tuning.so")


def needs_build(lib: str = LIB) -> bool:
    if not os.path.exists(lib):
        return True
    t = os.path.getmtime(lib)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + HEADERS
    return any(os.path.getmtime(d) > t for d in deps)


def build_tuning(force: bool = False, verbose: bool = False) -> str:
    """The -DKIVI_TUNING build (environment knobs, losing / diagnostic instantiations, fault injection): never loaded by the product
    path -- tools/ and tests/test_timeout_gpu.py select it with KIVI_TUNING=1 KIVI_HIP_LIB=<this file>."""
    return build(force, verbose, lib=TUNING_LIB, extra=["-DKIVI_TUNING", "-Wno-unused-value"], objname="_build_tuning")


def build(force: bool = False, verbose: bool = False, lib: str = LIB, extra=(), objname: str = "_build") -> str:
    if not force and not needs_build(lib):
        return lib
    LIB = lib                                          # noqa: N806 (the rest of the function writes `LIB`)
    os.makedirs(os.path.dirname(lib), exist_ok=True)
    hipcc = _hipcc()
    objdir = os.path.join(HERE, objname)
    os.makedirs(objdir, exist_ok=True)
    procs = []
    objs = []
    for src in SOURCES:
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        objs.append(obj)
        cmd = [hipcc, *FLAGS, *extra, "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    failed = False
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            failed = True
            sys.stderr.write(f"[kivi_amd.build] {src} failed:\n{out}\n")
        elif verbose and out.strip():
            print(out)
    if failed:
        raise RuntimeError("hipcc failed; see messages above")
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB + ".tmp", *objs]
    subprocess.run(cmd, check=True)
    os.replace(LIB + ".tmp", LIB)
    return LIB


if __name__ == "__main__":
    path = build(force="--force" in sys.argv, verbose=True)
    print("built", path)
    if "--tuning" in sys.argv:
        print("built", build_tuning(force="--force" in sys.argv, verbose=True))
