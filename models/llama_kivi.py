"""Import-path shim for the reference's models/llama_kivi.py attention hook."""
from kivi_amd.attention import (KiviConfig, KiviLayerCache, LlamaAttention_KIVI,  # noqa: F401
                                LlamaFlashAttention_KIVI, kivi_attention_decode, kivi_attention_prefill)

from kivi_amd.llama import LlamaForCausalLM_KIVI  # noqa: E402,F401  (decoder wrapper, models/llama_kivi.py:785)
