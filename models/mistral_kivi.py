"""Import-path shim for the reference's models/mistral_kivi.py (attention hook :69-534, decoder wrapper :921)."""
from kivi_amd.attention import (KiviConfig, KiviLayerCache, MistralAttention_KIVI,  # noqa: F401
                                MistralFlashAttention_KIVI, kivi_attention_decode, kivi_attention_prefill)
from kivi_amd.llama import MistralForCausalLM_KIVI  # noqa: F401
