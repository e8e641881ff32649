"""Import-path shim for the reference's models/mistral_kivi.py attention hook.

The reference's Mistral hook differs from the Llama one only in how it feeds GQA to the fused GEMV: it
materialises `repeat_kv_quant(...)` copies of the packed codes / scale / mn for every call
(mistral_kivi.py:58-67, :381-385, :441-445).  The kernels here map query heads onto kv heads themselves
(gemv_cuda.cu:361-365 semantics), so the same hook serves both families; sliding-window configs are passed
through untouched (the reference never applies the window to the quantised cache either).
"""
from kivi_amd.attention import (KiviConfig, KiviLayerCache, LlamaAttention_KIVI,  # noqa: F401
                                kivi_attention_decode, kivi_attention_prefill)

MistralAttention_KIVI = LlamaAttention_KIVI
MistralFlashAttention_KIVI = LlamaAttention_KIVI

from kivi_amd.llama import MistralForCausalLM_KIVI  # noqa: E402,F401  (decoder wrapper, models/mistral_kivi.py:921)
