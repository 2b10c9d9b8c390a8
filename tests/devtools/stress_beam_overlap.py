#!/usr/bin/env python3
"""Stress of the overlapped beam search (dev tool, GPU): python tests/devtools/stress_beam_overlap.py [seconds] [seed]

engine.forward_beam(overlap=True) -- what bench.py --config 4 times -- queues the search of batch k on a side stream behind its
log-probs and starts the acoustic pass of batch k + 1 on the main stream at once: two streams, two workspaces, an event each way, a
tensor allocated on one stream and last read on the other, the busy-CU hint that re-tiles the GEMMs.  Here a random sequence of
batches (1-70 ragged rows, 0.5-6 s, QuartzNet12x1 with a synthetic 3-gram LM in the .arpa behaviour, width 8-128) goes through it
back to back WITHOUT synchronising between steps; afterwards every result is compared, bit for bit, with the same batch through
overlap=False (one stream, fully serial).  A missing dependency shows as a difference.  Prints one JSON line."""
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import viet_asr_amd  # noqa: E402,F401
from viet_asr_amd import configs, synth  # noqa: E402
from viet_asr_amd.beam import BeamSearchDecoder  # noqa: E402
from viet_asr_amd.engine import QuartzNetCTC  # noqa: E402


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    model = sys.argv[3] if len(sys.argv) > 3 else "quartznet12x1_vi"      # quartznet15x5: BASELINE configs[3]'s own model and batch sizes
    cfg = configs.builtin(model)
    jas = cfg["JasperEncoder"]["jasper"]
    eng = QuartzNetCTC(cfg, synth.encoder_state_dict(jas, 64, 5), synth.decoder_state_dict(1024, len(cfg["labels"]) + 1, 5))
    arpa = os.path.join(tempfile.mkdtemp(prefix="vasr_stress_"), "lm.arpa")
    synth.synthetic_arpa(arpa, cfg["labels"], n_words=2000, n_bigrams=6000, n_trigrams=6000, seed=seed)
    dec = BeamSearchDecoder(cfg["labels"], lm_path=arpa, alpha=0.5, beta=1.5)
    rng = np.random.default_rng(seed)
    t0, rounds, batches, rows, diffs = time.time(), 0, 0, 0, []
    while time.time() - t0 < seconds:
        seq = []
        for _ in range(int(rng.integers(3, 12))):
            B = int(rng.choice([1, int(rng.integers(2, 16)), int(rng.integers(16, 71))]))
            L = int(rng.integers(8000, 96000)) if B < 16 else int(rng.integers(8000, 40000))
            if model == "quartznet15x5" and rng.random() < 0.5:
                B, L = 64, 160000                                                    # configs[3] itself
            sig, lens = synth.audio_batch(B, L, int(rng.integers(0, 1 << 30)), ragged=True)
            seq.append((torch.from_numpy(sig).cuda(), torch.from_numpy(lens).cuda(), int(rng.choice([8, 32, 128]))))
        torch.cuda.synchronize()
        out = [eng.forward_beam(w, n, dec, bw, overlap=True) for w, n, bw in seq]     # back to back, nothing waits
        torch.cuda.synchronize()
        got = [(r["ids"].clone(), r["id_len"].clone(), r["score"].clone()) for r in out]
        for k, (w, n, bw) in enumerate(seq):
            r = eng.forward_beam(w, n, dec, bw, overlap=False)
            torch.cuda.synchronize()
            ids, nn, sc = got[k]
            same = torch.equal(nn, r["id_len"]) and torch.equal(sc, r["score"]) and all(
                torch.equal(ids[b, : int(nn[b])], r["ids"][b, : int(nn[b])]) for b in range(ids.shape[0]))
            if not same:
                diffs.append((rounds, k, int(w.shape[0]), int(w.shape[1]), bw))
            batches += 1
            rows += int(w.shape[0])
        rounds += 1
    print(json.dumps({"model": model, "seconds": round(time.time() - t0, 1), "sequences": rounds, "batches": batches, "rows": rows, "differences": len(diffs), "first": diffs[:4]}))


if __name__ == "__main__":
    main()
