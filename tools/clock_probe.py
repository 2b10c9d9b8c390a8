#!/usr/bin/env python3
"""Effective-clock evidence for the matrix pipe under load (dev tool; run under
    rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d <dir> -- python tools/clock_probe.py
and divide GRBM_GUI_ACTIVE by the dispatch duration): the bare MFMA stream of the split GEMM (no loads, LDS or barriers;
vasr_bench_mfma_sustained, f16x2 stream) for 3 x ~25 ms.  Prints its own TFLOP/s for comparison."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import viet_asr_amd  # noqa
from viet_asr_amd import _lib
L = _lib.dev_lib()
sink = torch.zeros(16, device="cuda")
fl = ctypes.c_double()
n_cu = torch.cuda.get_device_properties(0).multi_processor_count
st = torch.cuda.current_stream().cuda_stream
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
run = lambda: _lib.check(L.vasr_bench_mfma_sustained(3, n_cu, steps, sink.data_ptr(), ctypes.byref(fl), st))
run(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(3): run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 3
print(f"bare MFMA stream: {fl.value / (ms * 1e-3) / 1e12:.1f} TFLOP/s f16 ({ms:.2f} ms per launch, {n_cu} workgroups x 8 wavefronts)")
