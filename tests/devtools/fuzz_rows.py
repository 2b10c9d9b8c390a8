#!/usr/bin/env python3
"""Randomised check of row-independent batching (dev tool, GPU; no oracle): python tests/devtools/fuzz_rows.py [n] [seed0] [max_seconds]

vasr_set_row_independent promises that a row of ANY batch comes out as the batch-1 call on that row alone would -- ids, id_len and
the log-probs of the row's own frames, BIT FOR BIT -- which is what the serving queue (serving.BatchingTranscriber, policy
"independent") and transcribe_manifest rely on.  The batch-1 call runs the batch <= 5 latency GEMM and the one-frame-per-wavefront
STFT, a batch of 6-70 rows the throughput tiles chosen by the tile rule: the promise is a statement about every pair of kernel
forms.  tests/test_gpu_round3.py pins it on one six-row batch per arithmetic; here: random architecture (or a shipped one), random
arithmetic, 2-70 ragged rows at different levels (float, or int16 PCM through the fused ingest), every row against its batch-1
call, plus the same rows in a shuffled batch of another size."""
import copy
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import viet_asr_amd  # noqa: E402,F401
from viet_asr_amd import configs, synth  # noqa: E402
from viet_asr_amd.engine import QuartzNetCTC  # noqa: E402
from test_gpu_parity import _random_architecture  # noqa: E402

STATS = {"cases": 0, "rows": 0, "pcm16_cases": 0}
_SHIPPED = {}


def rows_case(case):
    rng = np.random.default_rng(700000 + case)
    kind = int(rng.integers(0, 3))
    gemm = str(rng.choice(["f16x2", "f16x2", "bf16x3", "fp32"]))
    if kind == 0:                                             # a shipped model (engines cached per arithmetic)
        key = ("quartznet12x1_vi", gemm)
        if key not in _SHIPPED:
            cfg = configs.builtin(key[0])
            jas = cfg["JasperEncoder"]["jasper"]
            _SHIPPED[key] = QuartzNetCTC(cfg, synth.encoder_state_dict(jas, 64, 3), synth.decoder_state_dict(jas[-1]["filters"], len(cfg["labels"]) + 1, 3), gemm=gemm)
        eng = _SHIPPED[key]
    else:
        cfg = copy.deepcopy(configs.builtin("quartznet15x5"))
        jas = cfg["JasperEncoder"]["jasper"] = _random_architecture(rng)
        eng = QuartzNetCTC(cfg, synth.encoder_state_dict(jas, 64, case), synth.decoder_state_dict(jas[-1]["filters"], len(cfg["labels"]) + 1, case), gemm=gemm)
    B = int(rng.choice([int(rng.integers(2, 6)), int(rng.integers(6, 24)), int(rng.integers(24, 71))]))
    L = int(rng.integers(600, 9000)) if B > 24 else int(rng.integers(600, 30000))
    lens = rng.integers(300, L + 1, size=B).astype(np.int64)
    lens[int(rng.integers(0, B))] = L
    lens[int(rng.integers(0, B))] = int(rng.choice([300, 320, 321, 479, 480, 481]))
    lens[int(np.argmax(lens))] = L
    pcm = bool(rng.random() < 0.3)
    r = np.random.RandomState(case)
    if pcm:
        sig = np.zeros((B, L), dtype=np.int16)
        for b in range(B):
            sig[b, : lens[b]] = np.clip(r.randn(lens[b]) * float(rng.choice([40.0, 900.0, 9000.0])), -32768, 32767).astype(np.int16)
        STATS["pcm16_cases"] += 1
    else:
        sig = np.zeros((B, L), dtype=np.float32)
        for b in range(B):
            sig[b, : lens[b]] = (float(rng.choice([0.003, 0.05, 0.5])) * r.randn(lens[b])).astype(np.float32)
    dev = "cuda"
    full = eng.forward(torch.from_numpy(sig).to(dev), torch.from_numpy(lens).to(dev), want_logp=True, row_independent=True)
    perm = rng.permutation(B)[: max(1, int(rng.integers(1, B + 1)))]
    sub = eng.forward(torch.from_numpy(sig[perm][:, : int(lens[perm].max())].copy()).to(dev), torch.from_numpy(lens[perm].copy()).to(dev),
                      want_logp=True, row_independent=True)
    STATS["cases"] += 1
    STATS["rows"] += B
    check = list(rng.permutation(B)[:6]) + [int(np.argmin(lens))]
    for b in check:
        n = int(lens[b])
        one = eng.forward(torch.from_numpy(sig[b:b + 1, :n].copy()).to(dev), torch.from_numpy(lens[b:b + 1].copy()).to(dev),
                          want_logp=True, row_independent=True)
        k, f = int(one["id_len"][0]), one["logp"].shape[1]
        if int(full["id_len"][b]) != k or not torch.equal(full["ids"][b, :k], one["ids"][0, :k]):
            return f"rows case {case}: gemm {gemm} kind {kind} B {B} L {L} pcm {pcm}: ids of row {b} (len {n}) differ from its batch-1 call"
        if not torch.equal(full["logp"][b, :f], one["logp"][0]):
            return f"rows case {case}: gemm {gemm} kind {kind} B {B} L {L} pcm {pcm}: log-probs of row {b} (len {n}) differ from its batch-1 call"
    for j, b in enumerate(perm):
        k = int(full["id_len"][b])
        f = eng.frames(int(lens[b]))[1]
        if int(sub["id_len"][j]) != k or not torch.equal(sub["ids"][j, :k], full["ids"][b, :k]) or not torch.equal(sub["logp"][j, :f], full["logp"][b, :f]):
            return f"rows case {case}: gemm {gemm} kind {kind} B {B} L {L} pcm {pcm}: row {b} differs between a batch of {B} and a batch of {len(perm)}"
    return None


if __name__ == "__main__":
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    S0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    LIMIT = float(sys.argv[3]) if len(sys.argv) > 3 else 1e9
    t0, bad = time.time(), 0
    for case in range(S0, S0 + N):
        if time.time() - t0 > LIMIT:
            break
        msg = rows_case(case)
        if msg:
            bad += 1
            print("MISMATCH", msg, flush=True)
    print(f"{STATS['cases']} cases from {S0} ({STATS['rows']} rows, {STATS['pcm16_cases']} cases on int16 PCM), {bad} mismatches, {time.time() - t0:.0f} s")
