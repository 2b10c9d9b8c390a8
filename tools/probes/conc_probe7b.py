import ctypes as C, sys, os, threading, time, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import viet_asr_amd
from viet_asr_amd import configs, synth, stages, _lib
from viet_asr_amd.frontend_tables import frontend_description
L = _lib.dev_lib()
cfg = configs.builtin("quartznet12x1_vi")
pre = dict(cfg["AudioToMelSpectrogramPreprocessor"])
hraw = _lib.Handle(frontend=frontend_description(dict(pre, normalize=None))); hraw.finalize()
dev = torch.device("cuda:0")
def gemm_fn(cin, cout, B, T):
    ld = int(L.vasr_padded_frames(T))
    x = torch.relu(torch.randn(B, cin, ld, device=dev)); w = (torch.randn(cout, cin) / cin ** 0.5).contiguous()
    sc, sh = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
    y = torch.empty(B, cout, ld, device=dev)
    pk3 = torch.empty(cout * cin * 3, dtype=torch.int16); _lib.check(L.vasr_pack_pointwise_bf16x3(w.data_ptr(), cout, cin, cout, pk3.data_ptr()))
    w3 = pk3.to(dev)
    return lambda: _lib.check(L.vasr_bench_pointwise_bf16x3(x.data_ptr(), w3.data_ptr(), sc.data_ptr(), sh.data_ptr(), B, cin, cout, T, y.data_ptr(), torch.cuda.current_stream().cuda_stream))
big = torch.randn(32 << 20, device=dev)
lp = torch.randn(40, 29, 91, device=dev)
others = {
  "idle": lambda: time.sleep(0.0003),
  "bf16x3 GEMM 1024->128 B=40 T=29 (head-like)": gemm_fn(1024, 128, 40, 29),
  "bf16x3 GEMM 256->256 B=40 T=29": gemm_fn(256, 256, 40, 29),

}
for B, Lsamp in ((64, 160000),):
    sig, lens = synth.audio_batch(B, Lsamp, 53, ragged=True)
    w, n = torch.from_numpy(sig).cuda(), torch.from_numpy(lens).cuda()
    want = stages.melspec(hraw, w, n)[0].clone(); torch.cuda.synchronize()
    for name, other in others.items():
        stop = [False]; bad = [0]; calls = [0]
        def a():
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                while not stop[0]:
                    m, s = stages.melspec(hraw, w, n); st.synchronize(); calls[0] += 1
                    if not torch.equal(m, want): bad[0] += 1
        def b():
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                while not stop[0]:
                    other(); st.synchronize()
        ta, tb = threading.Thread(target=a), threading.Thread(target=b)
        ta.start(); tb.start(); time.sleep(5); stop[0] = True; ta.join(); tb.join()
        print(f"stft B={B} L={Lsamp} | other: {name:46s}: calls {calls[0]} wrong {bad[0]}", flush=True)
