"""Import-path shim for the reference's native module name (`import kivi_gemv`, quant/matmul.py:6)."""
from kivi_amd.quant.kivi_gemv import gemv_forward_cuda, gemv_forward_cuda_outer_dim  # noqa: F401
