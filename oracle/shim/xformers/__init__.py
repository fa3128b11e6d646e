"""Stand-in for xformers 0.0.26.post1 (pinned by the reference, poetry.lock:1927) -- TEST INFRASTRUCTURE ONLY.

The real package is not installable here (no network, CUDA-only wheels).  Only the four
attn_bias classes and `memory_efficient_attention` that the reference's hot path touches are
restated, in plain fp32 torch, from xformers' published semantics.  Used solely by
oracle/make_golden.py to run the UNMODIFIED reference on CPU; never imported by the product.
"""
