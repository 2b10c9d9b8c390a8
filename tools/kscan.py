import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import viet_asr_amd
from viet_asr_amd import _lib
L = _lib.lib(); dev = torch.device("cuda:0"); B, T = 64, 501; ld = 512
st = lambda: torch.cuda.current_stream().cuda_stream
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3
for cout in (512, 256):
  for cin in (128, 256, 512, 1024, 2048):
    x = torch.randn(B, cin, ld, device=dev); y = torch.empty(B, cout, ld, device=dev)
    wt = torch.randn(cin*cout, device=dev); sc, sh = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
    us = timeit(lambda: _lib.check(L.vasr_bench_pointwise(x.data_ptr(), wt.data_ptr(), sc.data_ptr(), sh.data_ptr(), B, cin, cout, T, y.data_ptr(), st())))
    print(f"M={cout} K={cin}: {us:8.1f} us  {2.0*cin*cout*B*ld/us/1e6:6.1f} TF(padded)")
