"""viet-asr_amd: MI355X-native (gfx950) hot path for dangvansam/viet-asr's infer.py.

mel front end -> QuartzNet encoder -> CTC head -> greedy / beam decode, as hand-written
HIP kernels behind a C-ABI library (csrc/, include/vasr.h) and the reference's
NeuralModule API (asr.py, core.py).
"""
from . import synth  # noqa: F401
