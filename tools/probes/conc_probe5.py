"""melspec under concurrency: what the other stream must run to corrupt it, and what the corruption looks like."""
import sys, os, threading, time, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import viet_asr_amd
from viet_asr_amd import configs, synth, stages, _lib
from viet_asr_amd.engine import QuartzNetCTC
cfg = configs.builtin("quartznet12x1_vi"); jas = cfg["JasperEncoder"]["jasper"]
enc_sd, dec_sd = synth.encoder_state_dict(jas, 64, 5), synth.decoder_state_dict(1024, 91, 5)
eng, eng2 = QuartzNetCTC(cfg, enc_sd, dec_sd, gemm="bf16x3"), QuartzNetCTC(cfg, enc_sd, dec_sd, gemm="bf16x3")
sig, lens = synth.audio_batch(40, 9000, 53, ragged=True)
w, n = torch.from_numpy(sig).cuda(), torch.from_numpy(lens).cuda()
w2, n2 = w.clone(), n.clone()
h, h2 = eng.handle, eng2.handle
mel, seq = stages.melspec(h, w, n)
enc2, _ = stages.encoder(h2, mel, seq, 1024)
torch.cuda.synchronize()
mel, seq, enc2 = mel.clone(), seq.clone(), enc2.clone()
first = [None]
def trial(name, other):
    stop = [False]; bad = [0]; calls = [0]; first[0] = None
    def a():
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            while not stop[0]:
                m, s = stages.melspec(h, w, n); st.synchronize(); calls[0] += 1
                if not torch.equal(m, mel):
                    bad[0] += 1
                    if first[0] is None:
                        d = (m != mel) | (torch.isnan(m) != torch.isnan(mel))
                        rows = torch.nonzero(d.any(2).any(1)).flatten().tolist()
                        b0 = rows[0]
                        fr = torch.nonzero(d[b0].any(0)).flatten().tolist()
                        bins = torch.nonzero(d[b0].any(1)).flatten().tolist()
                        first[0] = dict(n_diff=int(d.sum()), rows=rows[:10], seq_ok=bool(torch.equal(s, seq)), row=b0, frames=fr[:8], n_frames=len(fr), bins=len(bins),
                                        got=[round(float(v), 4) for v in m[b0, bins[0], fr[:4]]], want=[round(float(v), 4) for v in mel[b0, bins[0], fr[:4]]], seq_row=int(seq[b0]))
    def b():
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            while not stop[0]:
                other(); st.synchronize()
    ta, tb = threading.Thread(target=a), threading.Thread(target=b)
    ta.start(); tb.start(); time.sleep(6); stop[0] = True; ta.join(); tb.join()
    print(f"{name:40s}: calls {calls[0]} wrong {bad[0]} {first[0] or ''}", flush=True)
trial("other: idle", lambda: time.sleep(0.0005))
trial("other: melspec (own handle, same input)", lambda: stages.melspec(h2, w, n))
trial("other: melspec (own handle, own input)", lambda: stages.melspec(h2, w2, n2))
trial("other: encoder port entry", lambda: stages.encoder(h2, mel, seq, 1024))
trial("other: decoder port entry", lambda: stages.decoder(h2, enc2))
trial("other: fused forward, own input", lambda: eng2.forward(w2, n2, want_logp=True))
trial("other: fused forward, same input", lambda: eng2.forward(w, n, want_logp=True))
