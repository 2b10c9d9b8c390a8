import sys, ctypes, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import viet_asr_amd
from viet_asr_amd import _lib
L = _lib.dev_lib(); sink = torch.zeros(16, device="cuda"); fl = ctypes.c_double()
st = torch.cuda.current_stream().cuda_stream
MODE = int(sys.argv[1]) if len(sys.argv) > 1 else 3        # 3 = f16x2 (the default arithmetic's stream), 1 = bf16x3
run = lambda: _lib.check(L.vasr_bench_mfma_sustained(MODE, 256, 4000 * (2 if MODE == 3 else 1), sink.data_ptr(), ctypes.byref(fl), st))
run(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); [run() for _ in range(3)]; e1.record(); torch.cuda.synchronize()
print("sustained", round(3 * fl.value / (e0.elapsed_time(e1) * 1e-3) / 1e12, 1), "TFLOP/s")
