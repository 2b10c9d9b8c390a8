import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import viet_asr_amd
from viet_asr_amd import _lib
L = _lib.dev_lib(); dev = torch.device("cuda:0"); B, T = 64, 501; ld = 512
st = lambda: torch.cuda.current_stream().cuda_stream
cin, cout, mode = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
x = torch.randn(B, cin, ld, device=dev); y = torch.empty(B, cout, ld, device=dev)
w = (torch.randn(cout, cin) / cin ** 0.5).contiguous(); sc, sh = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
if mode == "fp32":
    pk = torch.empty(cout * cin); _lib.check(L.vasr_pack_pointwise(w.data_ptr(), cout, cin, cout, pk.data_ptr())); wt = pk.to(dev)
    fn = lambda: _lib.check(L.vasr_bench_pointwise(x.data_ptr(), wt.data_ptr(), sc.data_ptr(), sh.data_ptr(), B, cin, cout, T, y.data_ptr(), st()))
else:
    pk = torch.empty(cout * cin * 3, dtype=torch.int16); _lib.check(L.vasr_pack_pointwise_bf16x3(w.data_ptr(), cout, cin, cout, pk.data_ptr())); wt = pk.to(dev)
    fn = lambda: _lib.check(L.vasr_bench_pointwise_bf16x3(x.data_ptr(), wt.data_ptr(), sc.data_ptr(), sh.data_ptr(), B, cin, cout, T, y.data_ptr(), st()))
for _ in range(5): fn()
torch.cuda.synchronize()
