#!/usr/bin/env python3
"""Isolated-layer micro-benchmarks (dev tool): depthwise and pointwise kernels at the C3 shapes."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import viet_asr_amd  # noqa
from viet_asr_amd import _lib

L = _lib.dev_lib()
dev = torch.device("cuda:0")
B, T = 64, 501
ld = int(L.vasr_padded_frames(T))
st = lambda: torch.cuda.current_stream().cuda_stream


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3  # us


lens = torch.full((B,), T, dtype=torch.int32, device=dev)
print("depthwise (B=64, T=501):")
for C, K in ((256, 33), (256, 39), (512, 51), (512, 63), (512, 75)):
    x = torch.randn(B, C, ld, device=dev)
    y = torch.empty_like(x)
    w = torch.randn(C, K, device=dev)
    us = timeit(lambda: _lib.check(L.vasr_bench_depthwise(x.data_ptr(), w.data_ptr(), lens.data_ptr(), B, C, T, K, y.data_ptr(), st())))
    byt = 4.0 * 2 * B * C * T + 4 * C * K
    fl = 2.0 * K * C * B * T
    print(f"  C={C} K={K}: {us:8.1f} us  {byt / us / 1e3:7.1f} GB/s ({byt / us / 1e3 / 8000:.1%} of 8 TB/s)  {fl / us / 1e6:6.1f} TFLOP/s")
    # correctness spot check vs torch conv1d
    ref = torch.nn.functional.conv1d(x[:2, :, :T], w[:, None, :], padding=K // 2, groups=C)
    print(f"      max err vs torch {float((y[:2, :, :T] - ref).abs().max()):.2e}")
print("pointwise (B=64, T=501):")
for cin, cout in ((64, 256), (256, 256), (256, 512), (512, 512), (512, 1024), (2048, 512), (8192, 512), (2048, 256)):
    x = torch.randn(B, cin, ld, device=dev)
    y = torch.empty(B, cout, ld, device=dev)
    w = (torch.randn(cout, cin) / cin ** 0.5).contiguous()
    packed = torch.empty(cout * cin)
    _lib.check(L.vasr_pack_pointwise(w.data_ptr(), cout, cin, cout, packed.data_ptr()))
    wt, wdev = packed.to(dev), w.to(dev)
    sc, sh = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
    us = timeit(lambda: _lib.check(L.vasr_bench_pointwise(x.data_ptr(), wt.data_ptr(), sc.data_ptr(), sh.data_ptr(), B, cin, cout, T, y.data_ptr(), st())))
    fl = 2.0 * cin * cout * B * T
    print(f"  {cin}->{cout}: {us:8.1f} us  {fl / us / 1e6:6.1f} TFLOP/s ({fl / us / 1e6 / 157.3:.1%} of 157.3)")
    ref = torch.relu(torch.einsum("mk,bkt->bmt", wdev, x[:2, :, :T]))
    print(f"      max err vs torch {float((y[:2, :, :T] - ref).abs().max()):.2e}")
