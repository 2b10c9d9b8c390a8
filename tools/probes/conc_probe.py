import sys, os, threading, time, copy, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import viet_asr_amd
from viet_asr_amd import configs, synth, stages, _lib
from viet_asr_amd.engine import QuartzNetCTC
cfg = configs.builtin("quartznet12x1_vi"); jas = cfg["JasperEncoder"]["jasper"]
enc_sd, dec_sd = synth.encoder_state_dict(jas, 64, 5), synth.decoder_state_dict(1024, 91, 5)
eng = QuartzNetCTC(cfg, enc_sd, dec_sd)
pool = []
for i, (B, L) in enumerate([(1, 30000), (5, 20000), (14, 12000), (36, 9000), (2, 50000), (8, 16000)]):
    sig, lens = synth.audio_batch(B, L, 50 + i, ragged=True)
    pool.append((torch.from_numpy(sig).cuda(), torch.from_numpy(lens).cuda()))
def run_all(e):
    out = []
    for w, n in pool:
        r = e.forward(w, n, want_logp=True)
        out.append(r["logp"])
    return out
want = [x.clone() for x in run_all(eng)]; torch.cuda.synchronize()
def trial(name, other):
    stop = [False]; bad = [0]; calls = [0]
    def a():
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            while not stop[0]:
                for i, (w, n) in enumerate(pool):
                    r = eng.forward(w, n, want_logp=True); st.synchronize()
                    calls[0] += 1
                    if not torch.equal(r["logp"], want[i]): bad[0] += 1
    def b():
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            while not stop[0]:
                other(); st.synchronize()
    ta, tb = threading.Thread(target=a), threading.Thread(target=b)
    ta.start(); tb.start(); time.sleep(12); stop[0] = True; ta.join(); tb.join()
    print(name, "calls", calls[0], "wrong", bad[0], flush=True)
junk = torch.randn(2048, 2048, device="cuda")
def matmul():
    global junk
    junk = (junk @ junk).clamp_(-1, 1)
big = torch.randn(64 << 20, device="cuda")
def copyk():
    big.add_(1.0)
eng2 = QuartzNetCTC(cfg, enc_sd, dec_sd)
trial("other = idle python", lambda: time.sleep(0.001))
trial("other = torch matmul", matmul)
trial("other = torch elementwise 256 MB", copyk)
def other_fwd(i=[0]):
    w, n = pool[i[0] % len(pool)]; i[0] += 1
    eng2.forward(w, n, want_logp=True)
trial("other = second engine, forward", other_fwd)
hp = _lib.Handle(frontend=eng.frontend); hp.finalize()
def other_mel(i=[0]):
    w, n = pool[i[0] % len(pool)]; i[0] += 1
    stages.melspec(hp, w, n)
trial("other = front end only (own handle)", other_mel)
