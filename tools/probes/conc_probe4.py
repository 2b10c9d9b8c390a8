"""Which stage breaks under two-stream concurrency: thread A repeats ONE stage of engine `eng` on fixed inputs and checks its output,
thread B runs whole forwards of a second engine."""
import sys, os, threading, time, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import viet_asr_amd
from viet_asr_amd import configs, synth, stages, _lib
from viet_asr_amd.engine import QuartzNetCTC
gemm = sys.argv[1] if len(sys.argv) > 1 else "bf16x3"
cfg = configs.builtin("quartznet12x1_vi"); jas = cfg["JasperEncoder"]["jasper"]
enc_sd, dec_sd = synth.encoder_state_dict(jas, 64, 5), synth.decoder_state_dict(1024, 91, 5)
eng, eng2 = QuartzNetCTC(cfg, enc_sd, dec_sd, gemm=gemm), QuartzNetCTC(cfg, enc_sd, dec_sd, gemm=gemm)
sig, lens = synth.audio_batch(40, 9000, 53, ragged=True)
w, n = torch.from_numpy(sig).cuda(), torch.from_numpy(lens).cuda()
h = eng.handle
mel, seq = stages.melspec(h, w, n)
enc, enc_len = stages.encoder(h, mel, seq, 1024)
logp = stages.decoder(h, enc)
full = eng.forward(w, n, want_logp=True)["logp"].clone()
torch.cuda.synchronize()
mel, seq, enc, logp = mel.clone(), seq.clone(), enc.clone(), logp.clone()
def trial(name, fn, other=True):
    stop = [False]; bad = [0]; calls = [0]
    def a():
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            while not stop[0]:
                ok = fn(); st.synchronize(); calls[0] += 1
                if not ok: bad[0] += 1
    def b():
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            while not stop[0]:
                if other: eng2.forward(w, n, want_logp=True)
                st.synchronize()
    ta, tb = threading.Thread(target=a), threading.Thread(target=b)
    ta.start(); tb.start(); time.sleep(8); stop[0] = True; ta.join(); tb.join()
    print(f"[{gemm}] {name:34s}: calls {calls[0]} wrong {bad[0]}", flush=True)
trial("melspec", lambda: torch.equal(stages.melspec(h, w, n)[0], mel))
trial("encoder (port entry)", lambda: torch.equal(stages.encoder(h, mel, seq, 1024)[0], enc))
trial("decoder (port entry)", lambda: torch.equal(stages.decoder(h, enc), logp))
trial("fused forward", lambda: torch.equal(eng.forward(w, n, want_logp=True)["logp"], full))
trial("fused forward, other thread idle", lambda: torch.equal(eng.forward(w, n, want_logp=True)["logp"], full), other=False)
