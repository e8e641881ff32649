from kivi_amd.quant.new_pack import *  # noqa: F401,F403
from kivi_amd.quant.new_pack import __all__  # noqa: F401
