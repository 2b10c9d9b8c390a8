#!/usr/bin/env python3
"""How loose may the power-of-two scale of the 2 x fp16 operand split be?  (CPU study for DESIGN section 8, next step (a):
scales from an a-priori bound instead of the measured maximum.)  A K-deep dot product of ReLU-like activations with
weights, operands split as the kernels do (hi = rne16(s x), lo = rne16(s x - hi), three products), accumulated in fp64 so
that only the SPLIT's error shows; the scale puts `2^-k max|x|` instead of `max|x|` into [2^14, 2^15).
    python tools/split_scale_study.py"""
import numpy as np

rng = np.random.default_rng(0)
K, N = 512, 4096
x = np.maximum(rng.standard_normal((K, N)), 0.0) * np.exp(rng.standard_normal((K, 1)))   # channels at different levels
x[:, ::7] *= 1e-3                                                                          # quiet columns
w = rng.standard_normal((64, K)) / np.sqrt(K)
ref = w @ x


def split(v, s):
    hi = (v * s).astype(np.float16)
    lo = (v * s - hi.astype(np.float64)).astype(np.float16)
    return hi.astype(np.float64), lo.astype(np.float64)


sw = 2.0 ** (14 - np.floor(np.log2(np.abs(w).max())))
wh, wl = split(w, sw)
print("k (scale too small by 2^k)   max |err| / max |y|    median rel err of |y| > 1e-3 max")
for k in range(0, 15, 2):
    sx = 2.0 ** (14 - np.floor(np.log2(np.abs(x).max())) - k)
    xh, xl = split(x, sx)
    y = (wl @ xh + wh @ xl + wh @ xh) / (sw * sx)
    err = np.abs(y - ref)
    big = np.abs(ref) > 1e-3 * np.abs(ref).max()
    print(f"{k:2d}   {err.max() / np.abs(ref).max():.3e}   {np.median(err[big] / np.abs(ref[big])):.3e}")
y32 = (w.astype(np.float32) @ x.astype(np.float32)).astype(np.float64)
print("fp32 matmul for scale:", f"{np.abs(y32 - ref).max() / np.abs(ref).max():.3e}")
