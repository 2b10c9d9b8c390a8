#!/usr/bin/env python3
"""Soak of the two beam-search kernels against each other at serving sizes (dev tool, GPU):

    python tools/soak_beam.py [n_cases] [seed0] [repeats] [max_seconds]

For every case (28 / 90 / 127 labels + blank -- 128 classes is the kernels' limit --, 20-1 200 frames, width 8-128, posteriors
CTC-like / noisy / flat, with and without the synthetic 3-gram LM, ragged per-row frame counts) the same utterances are searched
  * as batches of 1, 3 and 15 rows: an utterance on four wavefronts (beam_group.hip), `repeats` times each, and
  * as a batch of 80 rows (the 16 utterances five times; beyond 64 rows): one wavefront per utterance (beam_wave.hip);
ids, lengths and scores must be the same bits everywhere (a race in the LDS-only barriers of the four-wavefront kernel would
show as a run-to-run difference, a scheduling-dependent merge order as a difference between the kernels).  No oracle here:
tests/test_beam.py and tests/test_gpu_configs.py compare both kernels with it.  Prints one JSON line."""
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


def posteriors(kind, rows, frames, labels, words, seed, r):
    V1 = len(labels) + 1
    if kind == 0:
        return synth.ctc_like_log_probs(rows, frames, labels, words, seed=seed, blank_frac=float(r.choice([0.3, 0.55, 0.8])),
                                        noise=float(r.choice([0.8, 1.3, 2.2])))
    z = r.randn(rows, frames, V1) * (float(r.choice([0.3, 1.0, 3.0])) if kind == 1 else 0.02)   # noisy / almost flat
    if kind == 1:
        z[..., V1 - 1] += float(r.choice([0.0, 2.0, 4.0]))
    z = z.astype(np.float64)
    return (z - np.log(np.exp(z).sum(-1, keepdims=True))).astype(np.float32)


def make_vocabs(tmp):
    vocabs = {}
    for name, labels in (("en29", list(" abcdefghijklmnopqrstuvwxyz'")), ("vi91", configs.builtin("quartznet12x1_vi")["labels"]),
                         ("wide128", [" "] + [chr(0x100 + i) for i in range(126)])):
        arpa = os.path.join(tmp, name + ".arpa")
        synth.synthetic_arpa(arpa, labels, seed=11)
        words = sorted(w[0] for w in read_arpa(arpa)[1] if len(w) == 1 and not w[0].startswith("<"))
        vocabs[name] = (labels, arpa, words)
    return vocabs


def run_case(seed, vocabs, decs, repeats, dev, stats):
    """None when every form and every repeat of case `seed` gave the same bits, else a description of the first difference."""
    r = np.random.RandomState(seed)
    name = ["en29", "vi91", "wide128"][r.randint(3)]
    labels, arpa, words = vocabs[name]
    use_lm = bool(r.randint(2))
    alpha, beta = float(r.choice([0.3, 0.5, 1.2])), float(r.choice([0.0, 1.5, 2.5]))
    frames = int(r.choice([20, 77, 140, 331, 501, 1200]))
    width = int(r.choice([8, 20, 50, 100, 128]))
    kind = int(r.randint(3))
    if kind == 2:
        frames = min(frames, 140)          # flat posteriors: every class a candidate on every frame, exact ties of scores
    rows = 16
    lp = torch.from_numpy(posteriors(kind, rows, frames, labels, words, seed, r)).to(dev)
    ragged = torch.from_numpy(r.randint(1, frames + 1, size=rows).astype(np.int32)) if r.randint(2) else None
    # with an LM: BOTH of pyctcdecode's behaviours (viet_asr_amd/beam.py) -- no unigram list ("binary"), unigram set + trie ("arpa")
    for mode in (("binary", "arpa") if use_lm else ("none",)):
        m = _run_mode(seed, name, mode, alpha, beta, labels, arpa, decs, lp, width, ragged, repeats, stats, frames, kind)
        if m:
            return m
    return None


def _run_mode(seed, name, mode, alpha, beta, labels, arpa, decs, lp, width, ragged, repeats, stats, frames, kind):
    key = (name, mode, alpha, beta)
    if key not in decs:
        decs[key] = BeamSearchDecoder(labels, lm_path=arpa if mode != "none" else None, alpha=alpha, beta=beta,
                                      unigrams=None if mode == "binary" else "auto")
    dec = decs[key]
    # the 16 utterances five times over = 80 rows: beyond 64 rows the library searches with one wavefront per utterance (beam_wave.hip)
    ref = [t.cpu()[:lp.shape[0]] for t in dec.decode_ids(lp.repeat(5, 1, 1), width, frames=ragged.repeat(5) if ragged is not None else None)]
    stats["searches"] += 1
    stats["overflow_rows"] += int((ref[1] < 0).sum())
    for lo, hi in ((0, 1), (1, 4), (1, 16)):                                     # 1, 3, 15 rows: beam_group.hip
        sub = lp[lo:hi].contiguous()
        fr = ragged[lo:hi] if ragged is not None else None
        for rep in range(repeats):
            ids, n, score = [t.cpu() for t in dec.decode_ids(sub, width, frames=fr)]
            stats["searches"] += 1
            stats["overflow_rows"] += int((n < 0).sum())
            for j in range(len(n)):
                k = int(n[j])
                if (k != int(ref[1][lo + j]) or not torch.equal(ids[j, :k], ref[0][lo + j, :k])
                        or float(score[j]).hex() != float(ref[2][lo + j]).hex()):
                    return {"case": seed, "vocab": name, "lm": mode, "frames": frames, "width": width, "kind": kind,
                            "batch_rows": [lo, hi], "row": lo + j, "repeat": rep, "ragged": ragged is not None,
                            "lengths": [k, int(ref[1][lo + j])], "scores": [float(score[j]), float(ref[2][lo + j])]}
    return None


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 500_000
    repeats = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    max_seconds = float(sys.argv[4]) if len(sys.argv) > 4 else 1e9
    dev = torch.device("cuda:0")
    vocabs = make_vocabs(tempfile.mkdtemp(prefix="vasr_soak_"))
    decs, bad, stats = {}, [], {"searches": 0, "overflow_rows": 0}
    t_start = time.time()
    done = 0
    for case in range(n_cases):
        if time.time() - t_start > max_seconds:
            break
        done += 1
        m = run_case(seed0 + case, vocabs, decs, repeats, dev, stats)
        if m:
            bad.append(m)
    print(json.dumps({"cases": done, "seed0": seed0, "repeats": repeats, **stats, "mismatches": len(bad), "first": bad[:5],
                      "seconds": round(time.time() - t_start, 1)}))
    return 1 if bad or stats["overflow_rows"] else 0


if __name__ == "__main__":
    sys.exit(main())
