import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import viet_asr_amd
from viet_asr_amd import _lib
L = _lib.lib(); dev = torch.device("cuda:0"); B, T = 64, 501; ld = 512
st = lambda: torch.cuda.current_stream().cuda_stream
cin, cout = int(sys.argv[1]), int(sys.argv[2])
x = torch.randn(B, cin, ld, device=dev); y = torch.empty(B, cout, ld, device=dev)
wt = torch.randn(cin*cout, device=dev); sc, sh = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
for _ in range(5):
    _lib.check(L.vasr_bench_pointwise(x.data_ptr(), wt.data_ptr(), sc.data_ptr(), sh.data_ptr(), B, cin, cout, T, y.data_ptr(), st()))
torch.cuda.synchronize()
