import ctypes as C, sys, os, threading, time, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import viet_asr_amd
from viet_asr_amd import configs, synth, stages, _lib
L = _lib.dev_lib(); dev = torch.device("cuda:0")
P = C.CDLL(os.path.join(os.path.dirname(_lib.LIB_PATH), "probe_lds_canary.so"))
P.lds_canary_launch.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
def gemm_fn(cin, cout, B, T, arith="bf16x3"):
    ld = int(L.vasr_padded_frames(T))
    x = torch.relu(torch.randn(B, cin, ld, device=dev)); w = (torch.randn(cout, cin) / cin ** 0.5).contiguous()
    sc, sh = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
    y = torch.empty(B, cout, ld, device=dev)
    pk3 = torch.empty(cout * cin * 3, dtype=torch.int16); _lib.check(L.vasr_pack_pointwise_bf16x3(w.data_ptr(), cout, cin, cout, pk3.data_ptr()))
    w3 = pk3.to(dev)
    return lambda: _lib.check(L.vasr_bench_pointwise_bf16x3(x.data_ptr(), w3.data_ptr(), sc.data_ptr(), sh.data_ptr(), B, cin, cout, T, y.data_ptr(), torch.cuda.current_stream().cuda_stream))
lp = torch.randn(40, 29, 91, device=dev)
others = {"idle": lambda: time.sleep(0.0003), "small bf16x3 GEMM 256->256 B=40 T=29": gemm_fn(256, 256, 40, 29), "head-like GEMM 1024->128 B=40 T=29": gemm_fn(1024, 128, 40, 29),
          "big GEMM 512->512 B=64 T=501": gemm_fn(512, 512, 64, 501), "torch log_softmax": lambda: torch.log_softmax(lp, -1)}
for name, other in others.items():
    for bytes_ in (56064, 16384):
        bad = torch.zeros(1, dtype=torch.int32, device=dev); first = torch.zeros(4, dtype=torch.int32, device=dev)
        stop = [False]; calls = [0]
        def a():
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                while not stop[0]:
                    P.lds_canary_launch(512, 512, bytes_, 40, bad.data_ptr(), first.data_ptr(), st.cuda_stream); st.synchronize(); calls[0] += 1
        def b():
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                while not stop[0]:
                    other(); st.synchronize()
        ta, tb = threading.Thread(target=a), threading.Thread(target=b)
        ta.start(); tb.start(); time.sleep(4); stop[0] = True; ta.join(); tb.join()
        torch.cuda.synchronize()
        f = first.cpu().numpy().astype(np.uint32)
        print(f"canary {bytes_:6d} B of LDS | other: {name:40s}: launches {calls[0]} changed words {int(bad.item())}" + (f" first: word {f[0]} = {f[1]:#010x} (pattern {f[2]:#010x}) block {f[3]}" if int(bad.item()) else ""), flush=True)
