#!/usr/bin/env python3
"""Time the device beam search on BASELINE config-4 shaped input (dev tool)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import viet_asr_amd
from viet_asr_amd import configs, synth
from viet_asr_amd.beam import BeamSearchDecoder
from viet_asr_amd.engine import QuartzNetCTC
cfg = configs.builtin("quartznet15x5"); jas = cfg["JasperEncoder"]["jasper"]
eng = QuartzNetCTC(cfg, synth.encoder_state_dict(jas, 64, 3), synth.decoder_state_dict(1024, 29, 3))
sig, lens = synth.audio_batch(64, 160000, 3)
r = eng.forward(torch.from_numpy(sig).cuda(), torch.from_numpy(lens).cuda(), want_logp=True)
logp = r["logp"]
for bw in (16, 128):
    dec = BeamSearchDecoder(cfg["labels"])
    dec.decode_ids(logp, bw); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): ids, n, sc = dec.decode_ids(logp, bw)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    print(f"beam_width {bw}: {dt*1e3:.1f} ms per batch of 64 x 501 frames; greedy-equal rows: "
          f"{sum(dec.decode_batch(logp[:4], bw)[i] == eng.texts(r['ids'][:4], r['id_len'][:4])[i] for i in range(4))}/4")

# with an n-gram LM (BASELINE config 4 shape): synthetic vocabulary, unigrams + bigrams, written as ARPA text
import tempfile
from oracle import beam_oracle as BO
rng = np.random.default_rng(4)
letters = [c for c in cfg["labels"] if c.strip() and c != "'"]
words = sorted({"".join(rng.choice(letters, rng.integers(2, 7))) for _ in range(3000)})
ng = {("<s>",): (-99.0, -0.3), ("</s>",): (-1.5, 0.0), ("<unk>",): (-3.0, 0.0)}
for w in words:
    ng[(w,)] = (float(-2.0 - 2.0 * rng.random()), float(-0.4 * rng.random()))
for _ in range(12000):
    a, b2 = rng.choice(words, 2)
    ng[(a, b2)] = (float(-0.5 - 2.0 * rng.random()), 0.0)
path = os.path.join(tempfile.mkdtemp(), "synthetic.arpa")
BO.write_arpa(path, 2, ng)
dec = BeamSearchDecoder(cfg["labels"], lm_path=path, alpha=0.5, beta=1.5)
for bw in (16, 128):
    dec.decode_ids(logp, bw); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): dec.decode_ids(logp, bw)
    torch.cuda.synchronize()
    print(f"beam_width {bw} + 2-gram LM ({len(words)} words): {(time.perf_counter() - t0) / 3 * 1e3:.1f} ms per batch of 64 x 501 frames")

# peaky posteriors (what a trained CTC model emits: one or two classes above token_min_logp per frame)
def peaky(T, V1, seed, k=4.0):
    r = np.random.RandomState(seed)
    z = r.randn(T, V1) * k
    z[:, -1] += 2.0; z[:, 0] += 1.0
    return (z - np.log(np.exp(z).sum(1, keepdims=True))).astype(np.float32)
lp = torch.from_numpy(np.stack([peaky(501, 29, b) for b in range(64)])).cuda()
for use_lm in (False, True):
    dec = BeamSearchDecoder(cfg["labels"], lm_path=path if use_lm else None, alpha=0.5, beta=1.5)
    for bw in (16, 128):
        dec.decode_ids(lp, bw); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3): dec.decode_ids(lp, bw)
        torch.cuda.synchronize()
        print(f"peaky posteriors, beam_width {bw}{' + LM' if use_lm else ''}: {(time.perf_counter() - t0) / 3 * 1e3:.1f} ms per batch of 64 x 501 frames")

# CTC-like posteriors: ~70 % of the frames are blank with p ~ 0.9999 (nothing else passes token_min_logp), the others carry
# one dominant character and a couple of alternatives -- what a converged CTC model emits
def ctc_like(T, V1, seed):
    r = np.random.RandomState(seed)
    z = r.randn(T, V1)
    blank = r.rand(T) < 0.7
    z[blank, -1] += 14.0
    idx = np.where(~blank)[0]
    z[idx, r.randint(0, V1 - 1, len(idx))] += 6.0
    return (z - np.log(np.exp(z).sum(1, keepdims=True))).astype(np.float32)
lp = torch.from_numpy(np.stack([ctc_like(501, 29, b) for b in range(64)])).cuda()
for use_lm in (False, True):
    dec = BeamSearchDecoder(cfg["labels"], lm_path=path if use_lm else None, alpha=0.5, beta=1.5)
    for bw in (16, 128):
        dec.decode_ids(lp, bw); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3): dec.decode_ids(lp, bw)
        torch.cuda.synchronize()
        print(f"CTC-like posteriors, beam_width {bw}{' + LM' if use_lm else ''}: {(time.perf_counter() - t0) / 3 * 1e3:.1f} ms per batch of 64 x 501 frames")
