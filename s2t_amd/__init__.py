"""s2t_amd — MI355X-native hot path of xuchennlp/S2T (speech-to-text fairseq fork).

Only the path named in BASELINE.json lives here: HIP/CDNA4 kernels behind a C-ABI library
(``csrc/`` -> ``lib/libs2t_hip.so``, declared in ``include/s2t_hip.h``) and the Python host side
that mirrors the reference's module / model interface.  There is no CPU fallback: using an op
without the built library (or without a GPU) raises.
"""
__version__ = "0.1.0"
