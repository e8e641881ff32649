"""Import-path shim: `import quant.new_pack`, `import quant.matmul` resolve to the MI355X implementation
(kivi_amd.quant) exactly where the reference keeps them (quant/new_pack.py, quant/matmul.py)."""
