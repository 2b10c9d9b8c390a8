#!/usr/bin/env python3
"""Beam search time per batch (64 x 501 frames) by beam width and posterior shape, merge-table size from VASR_BEAM_SLOTS (dev)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import viet_asr_amd
from viet_asr_amd import configs, synth
from viet_asr_amd.beam import BeamSearchDecoder
cfg = configs.builtin("quartznet15x5")
def peaky(T, V1, seed, k=4.0):
    r = np.random.RandomState(seed); z = r.randn(T, V1) * k; z[:, -1] += 2.0; z[:, 0] += 1.0
    return (z - np.log(np.exp(z).sum(1, keepdims=True))).astype(np.float32)
def flat(T, V1, seed):
    r = np.random.RandomState(seed); z = r.randn(T, V1) * 0.5
    return (z - np.log(np.exp(z).sum(1, keepdims=True))).astype(np.float32)
words = ["xin", "chao", "viet", "nam", "toi", "la", "mot", "hai", "ba", "bon"]
cases = {"flat": np.stack([flat(501, 29, b) for b in range(64)]), "peaky": np.stack([peaky(501, 29, b) for b in range(64)]),
         "ctc-like": synth.ctc_like_log_probs(64, 501, cfg["labels"], words, seed=4)}
dec = BeamSearchDecoder(cfg["labels"])
out = []
for name, lp in cases.items():
    lp = torch.from_numpy(lp).cuda()
    for bw in (8, 16, 20, 32, 50, 64, 100, 128):
        dec.decode_ids(lp, bw); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3): dec.decode_ids(lp, bw)
        torch.cuda.synchronize()
        out.append(f"{name}:{bw}={(time.perf_counter() - t0) / 3 * 1e3:.2f}")
print("slots", os.environ.get("VASR_BEAM_SLOTS", "auto"), " ".join(out), flush=True)
