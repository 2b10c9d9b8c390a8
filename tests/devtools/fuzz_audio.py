#!/usr/bin/env python3
"""Randomised device-vs-oracle checks of the resampler and of the greedy tail (argmax + CTC collapse) (dev tool)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import viet_asr_amd  # noqa
from viet_asr_amd import audio, stages
from oracle import audio_oracle as AO
from oracle import quartznet_oracle as O

RATES = [8000, 11025, 12000, 16000, 22050, 24000, 32000, 44100, 48000]


def resample_case(case):
    r = np.random.RandomState(20_000 + case)
    sr_in, sr_out = int(r.choice(RATES)), int(r.choice(RATES))
    if sr_in == sr_out:
        return None
    B = int(r.randint(1, 5))
    L = int(r.randint(40, 6000))
    lens = r.randint(max(1, L // 4), L + 1, size=B).astype(np.int64)
    lens[r.randint(B)] = L
    x = np.zeros((B, L), dtype=np.float32)
    for b in range(B):
        x[b, : lens[b]] = (0.3 * r.randn(lens[b])).astype(np.float32)
    y, ln = audio.resample(torch.from_numpy(x).cuda(), torch.from_numpy(lens).cuda(), sr_in, sr_out)
    y, ln = y.cpu().numpy(), ln.cpu().numpy()
    for b in range(B):
        ref = AO.resample(x[b, : lens[b]], sr_in, sr_out)
        if ln[b] != len(ref):
            return f"resample case {case} {sr_in}->{sr_out} B {B} L {L} row {b}: length {ln[b]} vs {len(ref)}"
        err = np.abs(y[b, : ln[b]] - ref).max() if len(ref) else 0.0
        if err > 4e-6 or y[b, ln[b]:].any():
            return f"resample case {case} {sr_in}->{sr_out} B {B} L {L} row {b}: max err {err:.2e}, tail nonzero {bool(y[b, ln[b]:].any())}"
    return None


def greedy_case(case):
    r = np.random.RandomState(30_000 + case)
    B, Tn, V1 = int(r.randint(1, 9)), int(r.randint(1, 700)), int(r.choice([2, 5, 29, 91, 128]))
    z = r.randn(B, Tn, V1).astype(np.float32)
    z[r.rand(B, Tn) < 0.5, V1 - 1] += 3.0                    # blank runs
    z = np.round(z * 4) / 4                                   # exact ties occur: lowest index must win
    pred = stages.greedy_argmax(torch.from_numpy(z).cuda())
    if not np.array_equal(pred.cpu().numpy(), z.argmax(-1)):
        return f"greedy case {case}: argmax differs"
    ids, n = stages.ctc_collapse(pred, V1 - 1)
    ids, n = ids.cpu().numpy(), n.cpu().numpy()
    for b in range(B):
        want = O.ctc_collapse_ids(z[b].argmax(-1), V1 - 1)
        if list(ids[b, : n[b]]) != list(want):
            return f"greedy case {case} row {b}: collapse differs"
    return None


if __name__ == "__main__":
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    S0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    LIMIT = float(sys.argv[3]) if len(sys.argv) > 3 else 1e9      # seconds per fuzzer: stop early, still print the summary
    t00, bad, done = time.time(), 0, []
    for fn in (resample_case, greedy_case):
        t0, k = time.time(), 0
        for case in range(S0, S0 + N):
            if time.time() - t0 > LIMIT:
                break
            msg = fn(case)
            k += 1
            if msg:
                bad += 1
                print("MISMATCH", msg, flush=True)
        done.append(k)
    print(f"{done} cases (resample, greedy) from {S0}, {bad} mismatches, {time.time() - t00:.0f} s")
