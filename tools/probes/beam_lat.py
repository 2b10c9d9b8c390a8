#!/usr/bin/env python3
"""Beam-search latency probe (dev): one utterance and a batch of 64, widths 20 / 50 / 100 / 128, with the bench's synthetic
3-gram LM, on CTC-like posteriors of 29 and 91 classes and on near-deterministic ("model-like": one class per frame, now and
then two) posteriors.  Library from VASR_LIB_PATH.  Prints one line per case: ms per call (median of 7)."""
import os, sys, tempfile, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import viet_asr_amd  # noqa: F401
from viet_asr_amd import configs, synth
from viet_asr_amd.beam import BeamSearchDecoder, read_arpa

frames = int(os.environ.get("FRAMES", "501"))
widths = [int(w) for w in os.environ.get("WIDTHS", "20,50,100,128").split(",")]
batches = [int(b) for b in os.environ.get("BATCHES", "1,64").split(",")]


def model_like(B, T, V1, seed):
    r = np.random.RandomState(seed)
    z = r.randn(B, T, V1).astype(np.float32)
    top = r.randint(0, V1, (B, T))
    np.put_along_axis(z, top[..., None], 12.0, -1)
    second = r.rand(B, T) < 0.1
    alt = r.randint(0, V1, (B, T))
    zz = np.take_along_axis(z, alt[..., None], -1)[..., 0]
    zz[second] = 10.0
    np.put_along_axis(z, alt[..., None], zz[..., None], -1)
    return (z - np.log(np.exp(z).sum(-1, keepdims=True))).astype(np.float32)


def med(fn, n=7):
    if os.environ.get("ONCE"):          # profile builds print one line per launch: a single call per case
        fn(); torch.cuda.synchronize(); print(flush=True); return 0.0
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return float(np.median(ts)) * 1e3


out = []
for model in ("quartznet15x5", "quartznet12x1_vi"):
    cfg = configs.builtin(model)
    labels = cfg["labels"]
    arpa = os.path.join(tempfile.mkdtemp(prefix="vasr_lm_"), "s3.arpa")
    synth.synthetic_arpa(arpa, labels, seed=3)
    words = sorted(w[0] for w in read_arpa(arpa)[1] if len(w) == 1 and not w[0].startswith("<"))
    dec = BeamSearchDecoder(labels, lm_path=arpa, alpha=0.5, beta=1.5)
    nolm = BeamSearchDecoder(labels, lm_path=None)
    V1 = len(labels) + 1
    for B in batches:
        cases = {"ctc": synth.ctc_like_log_probs(B, frames, labels, words, seed=5), "model": model_like(B, frames, V1, 5)}
        for name, lp in cases.items():
            lp = torch.from_numpy(lp).cuda()
            row = [f"{model[9:]}/V{V1}/B{B}/{name}:"]
            for w in widths:
                row.append(f"w{w}={med(lambda: dec.decode_ids(lp, w)):.2f}")
            row.append(f"nolm w{widths[-2]}={med(lambda: nolm.decode_ids(lp, widths[-2])):.2f}")
            out.append(" ".join(row))
            print(out[-1], flush=True)
