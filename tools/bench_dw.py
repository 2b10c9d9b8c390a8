#!/usr/bin/env python3
"""Isolated depthwise layer timing (dev tool): Toeplitz / MFMA kernel vs the packed-FMA kernel, B x C x T like the bench.
    python tools/bench_dw.py [K ...]           VASR_LIB_PATH selects an alternative build (ablations)"""
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
for K in [int(a) for a in sys.argv[1:]] or [33, 51, 75]:
    Cn = 256 if K < 51 else 512
    dil = 2 if K == 87 else 1
    x = torch.randn(B, Cn, ld, device=dev); w = torch.randn(Cn, K) / K ** 0.5
    y = torch.empty_like(x); lens = torch.full((B,), T, dtype=torch.int32, device=dev)
    tsz = int(L.vasr_depthwise_mfma_table_size(K, dil))
    tab, inv = torch.empty(Cn, tsz, dtype=torch.int32), torch.empty(Cn)
    _lib.check(L.vasr_pack_depthwise_taps(w.data_ptr(), Cn, K, dil, tab.data_ptr(), inv.data_ptr()))
    tab, inv, wd = tab.to(dev), inv.to(dev), w.to(dev)
    stride = max(256, Cn * ((ld + 255) // 256) * 4)
    amax = torch.zeros(2, B, stride, dtype=torch.int32, device=dev)
    # the bench entry also runs the maxima pre-pass and two memsets: time those alone and subtract
    t_m = timeit(lambda: _lib.check(L.vasr_bench_depthwise_mfma(x.data_ptr(), tab.data_ptr(), inv.data_ptr(), lens.data_ptr(), B, Cn, T, K, dil, y.data_ptr(), amax.data_ptr(), stride, st())))
    t_v = timeit(lambda: _lib.check(L.vasr_bench_depthwise(x.data_ptr(), wd.data_ptr(), lens.data_ptr(), B, Cn, T, K, y.data_ptr(), st()))) if dil == 1 else float("nan")
    t_c = timeit(lambda: y.copy_(x))
    mb = 2 * x.numel() * 4 / 1e6
    print(f"K={K} C={Cn} B={B} T={T}: toeplitz(+amax pre-pass) {t_m:.1f} us, packed-FMA {t_v:.1f} us, copy_ {t_c:.1f} us ({mb:.0f} MB -> {mb / t_c:.2f} TB/s)", flush=True)
