#!/usr/bin/env python3
"""HBM copy ceiling at the sizes the depthwise kernels work on (dev tool): builds tools/probes/hbm_probe.hip on the box."""
import ctypes as C, os, subprocess, sys, torch
here = os.path.dirname(os.path.abspath(__file__))
so = "/tmp/hbm_probe.so"
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", os.path.join(here, "hbm_probe.hip"), "-o", so], check=True)
L = C.CDLL(so); L.probe.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]
dev = torch.device("cuda:0")
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3
for mb in (134, 403, 1073):
    n = mb * 1000 * 1000 // 2 // 16 * 16 // 4     # floats per tensor: read + write = mb MB
    x = torch.randn(n, device=dev); y = torch.empty_like(x)
    st = torch.cuda.current_stream().cuda_stream
    us = timeit(lambda: y.copy_(x))
    print(f"--- {mb} MB (r+w): torch copy_ {us:.1f} us = {2*n*4/us/1e3:.0f} GB/s")
    names = {0: "1xf4", 1: "4xf4", 2: "8xf4", 3: "persist4", 4: "persist8"}
    for kind in range(5):
        for mode, mn in ((0, "plain"), (1, "nt ld+st"), (2, "nt st"), (3, "nt ld")):
            for grid in ((256 * 4, 256 * 8, 256 * 16) if kind >= 3 else (0,)):
                r = L.probe(kind, mode, x.data_ptr(), y.data_ptr(), n // 4, grid, st)
                if r != 0: continue
                us = timeit(lambda: L.probe(kind, mode, x.data_ptr(), y.data_ptr(), n // 4, grid, st))
                assert torch.equal(x, y)
                print(f"  {names[kind]:9s} {mn:9s} grid {grid:5d}: {us:7.1f} us = {2*n*4/us/1e3:6.0f} GB/s")
