"""Two host threads, two streams, each launching isolated split-GEMM layers (devtools entry) on buffers of its own: (a) ONE launch per
check, (b) a chain of dependent launches per check.  Results against the same launches run alone."""
import ctypes as C, os, sys, threading, time
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import viet_asr_amd  # noqa
from viet_asr_amd import _lib
L = _lib.dev_lib()
dev = torch.device("cuda:0")
def make(cin, cout, B, T, seed):
    g = torch.Generator().manual_seed(seed)
    ld = int(L.vasr_padded_frames(T))
    x = torch.relu(torch.randn(B, cin, ld, generator=g)).to(dev)
    w = (torch.randn(cout, cin, generator=g) / cin ** 0.5).contiguous()
    sc, sh = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
    pk3 = torch.empty(cout * cin * 3, dtype=torch.int16); _lib.check(L.vasr_pack_pointwise_bf16x3(w.data_ptr(), cout, cin, cout, pk3.data_ptr()))
    return dict(cin=cin, cout=cout, B=B, T=T, ld=ld, x=x, w3=pk3.to(dev), sc=sc, sh=sh)
def gemm(p, x, y):
    _lib.check(L.vasr_bench_pointwise_bf16x3(x.data_ptr(), p["w3"].data_ptr(), p["sc"].data_ptr(), p["sh"].data_ptr(), p["B"], p["cin"], p["cout"], p["T"],
                                             y.data_ptr(), torch.cuda.current_stream().cuda_stream))
def run(p, chain, bufs):
    src = p["x"]
    for i in range(chain):
        gemm(p, src, bufs[i % 2]); src = bufs[i % 2]
    return src
def trial(cin, B, T, chain, secs=6):
    ps = [make(cin, cin, B, T, 1), make(cin, cin, B, T, 2)]
    bufs = [[torch.empty(B, cin, p["ld"], device=dev) for _ in range(2)] for p in ps]
    want = []
    for p, b in zip(ps, bufs):
        want.append(run(p, chain, b).clone()); torch.cuda.synchronize()
    stop = [False]; bad = [0, 0]; calls = [0, 0]
    def worker(k):
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            while not stop[0]:
                out = run(ps[k], chain, bufs[k]); st.synchronize(); calls[k] += 1
                if not torch.equal(out, want[k]): bad[k] += 1
    ths = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
    for t in ths: t.start()
    time.sleep(secs); stop[0] = True
    for t in ths: t.join()
    print(f"{cin}->{cin} B={B} T={T} chain {chain:2d}: calls {calls} wrong {bad}", flush=True)
for cin, B, T in ((256, 40, 57), (256, 14, 76), (512, 40, 57), (256, 64, 501)):
    for chain in (1, 2, 12):
        trial(cin, B, T, chain)
