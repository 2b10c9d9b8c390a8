#!/usr/bin/env python3
"""Depthwise kernel time against the number of utterance pairs (dev tool): does the workgroup count per CU quantise?"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import viet_asr_amd  # noqa
from viet_asr_amd import _lib
L = _lib.dev_lib(); dev = torch.device("cuda:0"); T = 501
ld = int(L.vasr_padded_frames(T)); st = lambda: torch.cuda.current_stream().cuda_stream
def timeit(fn, iters=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3
for C, K in ((256, 33), (512, 51), (512, 75)):
    for B in (40, 48, 56, 64, 72, 80, 96, 112, 128):
        x = torch.randn(B, C, ld, device=dev); y = torch.empty_like(x); w = torch.randn(C, K, device=dev)
        lens = torch.full((B,), T, dtype=torch.int32, device=dev)
        us = timeit(lambda: _lib.check(L.vasr_bench_depthwise(x.data_ptr(), w.data_ptr(), lens.data_ptr(), B, C, T, K, y.data_ptr(), st())))
        byt = 8.0 * B * C * T
        wgs = (C // 4) * ((B + 1) // 2)
        print(f"C={C} K={K} B={B:3d}: {us:7.1f} us  {byt / us / 1e3:7.0f} GB/s  workgroups/CU {wgs / 256:5.2f}  us per wg/CU {us / (wgs / 256):.2f}", flush=True)
