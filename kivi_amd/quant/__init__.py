"""Drop-in for the reference's quant/ package (new_pack, matmul, kivi_gemv)."""
