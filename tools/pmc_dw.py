import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import viet_asr_amd
from viet_asr_amd import _lib
L = _lib.dev_lib(); dev = torch.device("cuda:0"); B, T = 64, 501; ld = 512
C, K = int(sys.argv[1]), int(sys.argv[2])
st = lambda: torch.cuda.current_stream().cuda_stream
lens = torch.full((B,), T, dtype=torch.int32, device=dev)
x = torch.randn(B, C, ld, device=dev); y = torch.empty_like(x); w = torch.randn(C, K, device=dev)
for _ in range(5):
    _lib.check(L.vasr_bench_depthwise(x.data_ptr(), w.data_ptr(), lens.data_ptr(), B, C, T, K, y.data_ptr(), st()))
torch.cuda.synchronize()
