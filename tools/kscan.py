import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import viet_asr_amd
from viet_asr_amd import _lib
L = _lib.dev_lib(); dev = torch.device("cuda:0"); B, T = 64, 501; ld = 512
st = lambda: torch.cuda.current_stream().cuda_stream
mode = sys.argv[1] if len(sys.argv) > 1 else "fp32"
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3
for cout in (512, 256):
  for cin in (128, 256, 512, 1024, 2048, 8192):
    x = torch.randn(B, cin, ld, device=dev); y = torch.empty(B, cout, ld, device=dev)
    w = (torch.randn(cout, cin) / cin ** 0.5).contiguous()
    sc, sh = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
    if mode == "fp32":
        pk = torch.empty(cout * cin); _lib.check(L.vasr_pack_pointwise(w.data_ptr(), cout, cin, cout, pk.data_ptr())); wt = pk.to(dev)
        fn = lambda: _lib.check(L.vasr_bench_pointwise(x.data_ptr(), wt.data_ptr(), sc.data_ptr(), sh.data_ptr(), B, cin, cout, T, y.data_ptr(), st()))
    else:
        pk = torch.empty(cout * cin * 3, dtype=torch.int16); _lib.check(L.vasr_pack_pointwise_bf16x3(w.data_ptr(), cout, cin, cout, pk.data_ptr())); wt = pk.to(dev)
        fn = lambda: _lib.check(L.vasr_bench_pointwise_bf16x3(x.data_ptr(), wt.data_ptr(), sc.data_ptr(), sh.data_ptr(), B, cin, cout, T, y.data_ptr(), st()))
    us = timeit(fn)
    ref = torch.relu(torch.einsum("mk,bkt->bmt", w.double().to(dev), x[:1, :, :T].double()))
    err = float((y[:1, :, :T].double() - ref).abs().max())
    print(f"{mode} M={cout} K={cin}: {us:8.1f} us  {2.0*cin*cout*B*ld/us/1e6:6.1f} TF-equiv(padded)  max err vs fp64 {err:.2e}")
