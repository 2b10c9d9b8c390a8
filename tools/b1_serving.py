#!/usr/bin/env python3
"""The reference's own serving shape under a profiler: batch 1, QuartzNet12x1 (Vietnamese head), greedy and beam search
(width 50 / 100, 3-gram LM) -- infer.py:181-192, app.py:22-28.  Run under `rocprofv3 --kernel-trace --stats` for the per-kernel
table of a batch-1 call; prints wall latencies itself.

    python tools/b1_serving.py [--model quartznet12x1_vi] [--seconds 6.6] [--calls 50]"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import viet_asr_amd  # noqa: E402,F401
from viet_asr_amd import configs, synth  # noqa: E402
from viet_asr_amd.beam import BeamSearchDecoder, read_arpa  # noqa: E402
from viet_asr_amd.engine import QuartzNetCTC  # noqa: E402


def lat(fn, n):
    for _ in range(5):
        fn()
    ts = []
    for _ in range(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return round(float(np.median(ts)) * 1e3, 4)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="quartznet12x1_vi")
    ap.add_argument("--seconds", type=float, default=6.6)
    ap.add_argument("--calls", type=int, default=50)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--no-beam", action="store_true")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    cfg = configs.builtin(a.model)
    jas = cfg["JasperEncoder"]["jasper"]
    eng = QuartzNetCTC(cfg, synth.encoder_state_dict(jas, 64, 3), synth.decoder_state_dict(jas[-1]["filters"], len(cfg["labels"]) + 1, 3))
    sig, lens = synth.audio_batch(a.batch, int(a.seconds * 16000), 5)
    wav, ln = torch.from_numpy(sig).to(dev), torch.from_numpy(lens).to(dev)
    out = {"model": a.model, "batch": a.batch, "seconds": a.seconds}
    out["greedy_ms"] = lat(lambda: eng.forward(wav, ln, want_logp=False, want_pred=False), a.calls)
    out["acoustic_logp_ms"] = lat(lambda: eng.forward(wav, ln, want_logp=True, want_pred=False), a.calls)
    if not a.no_beam:
        arpa = os.path.join(tempfile.mkdtemp(prefix="vasr_lm_"), "synthetic3.arpa")
        synth.synthetic_arpa(arpa, cfg["labels"], seed=3)
        words = sorted(w[0] for w in read_arpa(arpa)[1] if len(w) == 1 and not w[0].startswith("<"))
        lp = eng.forward(wav, ln, want_logp=True, want_pred=False)["logp"]
        lp_ctc = torch.from_numpy(synth.ctc_like_log_probs(a.batch, lp.shape[1], cfg["labels"], words, seed=5)).to(dev)
        # both of pyctcdecode's LM behaviours (viet_asr_amd/beam.py): no unigram list ("binary"), unigram set + trie ("arpa")
        for mode in ("binary", "arpa"):
            dec = BeamSearchDecoder(cfg["labels"], lm_path=arpa, alpha=0.5, beta=1.5, unigrams=None if mode == "binary" else "auto")
            for width in (20, 50, 100, 128):
                out[f"beam{width}_search_model_{mode}_ms"] = lat(lambda: dec.decode_ids(lp, width), max(10, a.calls // 3))
                out[f"beam{width}_search_ctc_like_{mode}_ms"] = lat(lambda: dec.decode_ids(lp_ctc, width), max(10, a.calls // 3))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
