"""kivi_amd -- MI355X (gfx950) implementation of KIVI's quant/ hot path.

Host side is Python on PyTorch-ROCm (device memory + streams only); all compute
is hand-written HIP behind the C ABI in include/kivi_hip.h (libkivi_hip.so).
There is no CPU fallback: every op raises if the library is missing or a tensor
is not on the GPU.
"""
__version__ = "0.1.0"
