"""MI355X-native implementation of mistral-inference's Transformer.forward_partial hot path.

Same import paths and names as the reference package (mistralai/mistral-inference 1.6.0) for the
surfaces on that path; all device work is done by libmistral_hip.so (hand-written gfx950 kernels).
"""
__version__ = "1.6.0+mi355x.1"
