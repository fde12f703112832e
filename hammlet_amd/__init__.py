"""hammlet_amd - MI355X-native Forward-Backward Gibbs sampling over wavelet-compressed blocks.

The product is the gfx950 shared library behind include/hml.h (hammlet_amd/csrc) and the `hammlet`
command-line driver; this package is the thin Python mirror of that C ABI used by the tests, the
benchmark and the multi-GPU chain pooling.
"""
from .capi import allreduce_marginals, iterate_many, Chain, debug_eval, Pool, HmlError, load_library, marginals_text, parse_text, synth_depth, synth_gauss  # noqa: F401
from . import build  # noqa: F401
