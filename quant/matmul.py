from kivi_amd.quant.matmul import *  # noqa: F401,F403
from kivi_amd.quant.matmul import __all__  # noqa: F401
