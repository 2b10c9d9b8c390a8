#!/usr/bin/env python3
"""Isolated 1x1-conv GEMM layer timing (dev tool), all arithmetics:  python tools/bench_pw.py [cin cout] ...
VASR_LIB_PATH selects an alternative build (ablations: results wrong, timing only)."""
import ctypes as C, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import viet_asr_amd  # noqa
from viet_asr_amd import _lib
L = _lib.dev_lib()
dev = torch.device("cuda:0")
B, T = int(os.environ.get("B", 64)), int(os.environ.get("T", 501))
ld = int(L.vasr_padded_frames(T))
st = lambda: torch.cuda.current_stream().cuda_stream
def timeit(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
args = [int(a) for a in sys.argv[1:]] or [512, 512, 256, 256]
for cin, cout in zip(args[::2], args[1::2]):
    x = torch.relu(torch.randn(B, cin, ld, device=dev)); w = (torch.randn(cout, cin) / cin ** 0.5).contiguous()
    sc, sh = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
    y = torch.empty(B, cout, ld, device=dev)
    pk3 = torch.empty(cout * cin * 3, dtype=torch.int16); _lib.check(L.vasr_pack_pointwise_bf16x3(w.data_ptr(), cout, cin, cout, pk3.data_ptr()))
    pk16 = torch.empty(cout * cin * 2, dtype=torch.int16); inv = C.c_float()
    _lib.check(L.vasr_pack_pointwise_f16x2(w.data_ptr(), cout, cin, cout, pk16.data_ptr(), C.byref(inv)))
    w3, w16 = pk3.to(dev), pk16.to(dev)
    stride = 1024
    amax = torch.zeros(2, B, stride, dtype=torch.int32, device=dev)
    t16 = timeit(lambda: _lib.check(L.vasr_bench_pointwise_f16x2(x.data_ptr(), w16.data_ptr(), inv.value, sc.data_ptr(), sh.data_ptr(), B, cin, cout, T, y.data_ptr(), amax.data_ptr(), stride, st())))
    t3 = timeit(lambda: _lib.check(L.vasr_bench_pointwise_bf16x3(x.data_ptr(), w3.data_ptr(), sc.data_ptr(), sh.data_ptr(), B, cin, cout, T, y.data_ptr(), st())))
    fl = 2.0 * cin * cout * B * T
    print(f"{cin}->{cout} B={B} T={T}: f16x2 {t16:.1f} us ({3 * fl / t16 / 1e6:.0f} TF executed), bf16x3 {t3:.1f} us ({6 * fl / t3 / 1e6:.0f} TF executed)", flush=True)
