import ctypes as C, sys, os, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import viet_asr_amd
from viet_asr_amd import _lib
L = _lib.dev_lib(); dev = torch.device("cuda:0")
def check(cin, cout, B, T):
    ld = int(L.vasr_padded_frames(T))
    x = torch.relu(torch.randn(B, cin, ld, device=dev)); w = (torch.randn(cout, cin) / cin ** 0.5).contiguous()
    sc, sh = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
    pk3 = torch.empty(cout * cin * 3, dtype=torch.int16); _lib.check(L.vasr_pack_pointwise_bf16x3(w.data_ptr(), cout, cin, cout, pk3.data_ptr()))
    w3 = pk3.to(dev)
    n = B * cout * ld
    guard = 1 << 22
    arena = torch.full((guard + n + guard,), 12345.0, device=dev)
    y = arena[guard:guard + n]
    _lib.check(L.vasr_bench_pointwise_bf16x3(x.data_ptr(), w3.data_ptr(), sc.data_ptr(), sh.data_ptr(), B, cin, cout, T, y.data_ptr(), torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    lo = int((arena[:guard] != 12345.0).sum()); hi = int((arena[guard + n:] != 12345.0).sum())
    inside_untouched = int((y == 12345.0).sum())
    print(f"GEMM {cin}->{cout} B={B} T={T} ld={ld}: words changed below y {lo}, above y {hi}; words of y left untouched {inside_untouched} of {n}", flush=True)
for shp in ((256, 256, 40, 29), (1024, 128, 40, 29), (256, 256, 40, 57), (512, 512, 64, 501), (256, 256, 1, 29), (1024, 128, 3, 501)):
    check(*shp)
