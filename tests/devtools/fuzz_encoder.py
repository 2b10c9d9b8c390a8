#!/usr/bin/env python3
"""Randomised whole-path comparison against the oracle over ARCHITECTURES and BATCH SHAPES (dev tool, GPU):

    python tests/devtools/fuzz_encoder.py [n_cases] [seed0] [max_seconds]

tests/test_gpu_parity.py::test_random_architectures_and_shapes_match_oracle draws batches of 1-5 rows, i.e. only ever reaches the
batch <= 5 latency GEMM.  Here the batch is drawn from 1-5 / 6-20 / 21-72 rows of short clips (the oracle stays cheap), so that the
product library's own kernel choice -- the 512 x 128 / 256 x 128 / 256 x 64 / 128 x 64 GEMM tile rule, the fused depthwise +
pointwise kernel on 128- and 64-frame tiles where it "fills the chip", the Toeplitz depthwise with its utterances-per-wavefront
rule, dual-K residual GEMMs -- is exercised on block lists the shipped models do not have (128-512 filters, kernels 3-99, repeats
1-3, dilation, strided prologue), in a randomly drawn GEMM arithmetic, on ragged batches with one very short row.
Checks: log-probs within the goldens' tolerance, encoded lengths equal, predictions equal wherever the oracle's margin exceeds twice
the tolerance, everything finite.  Prints one summary line."""
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
from oracle import quartznet_oracle as O  # noqa: E402  (checker only)
from test_gpu_parity import _random_architecture  # noqa: E402

STATS = {"cases": 0, "rows": 0, "worst_err_over_tol": 0.0, "by_batch_class": [0, 0, 0], "fused_launches": 0}


def encoder_case(case):
    rng = np.random.default_rng(900000 + case)
    cfg = copy.deepcopy(configs.builtin("quartznet15x5"))
    jas = cfg["JasperEncoder"]["jasper"] = _random_architecture(rng)
    if rng.random() < 0.5:      # a stack of 256-channel K = 33 / 39 sub-blocks: the fused kernel's shapes, in the PRODUCT's own choice
        jas.insert(1, dict(filters=256, repeat=int(rng.integers(1, 4)), kernel=[int(rng.choice([33, 39]))], stride=[1], dilation=[1],
                           dropout=0.0, residual=bool(rng.random() < 0.7), separable=True))
        jas.insert(1, dict(filters=256, repeat=1, kernel=[33], stride=[1], dilation=[1], dropout=0.0, residual=False, separable=True))
    enc_sd = synth.encoder_state_dict(jas, 64, case)
    dec_sd = synth.decoder_state_dict(jas[-1]["filters"], len(cfg["labels"]) + 1, case)
    gemm = str(rng.choice(["f16x2", "f16x2", "bf16x3", "fp32"]))
    eng = QuartzNetCTC(cfg, enc_sd, dec_sd, gemm=gemm)
    cls = int(rng.integers(0, 3))
    B = int(rng.integers(1, 6)) if cls == 0 else int(rng.integers(6, 21)) if cls == 1 else int(rng.integers(21, 73))
    L = int(rng.integers(1500, 30000)) if cls == 0 else int(rng.integers(1500, 12000)) if cls == 1 else int(rng.integers(1500, 6000))
    sig, lens = synth.audio_batch(B, L, case, ragged=True)
    lens[int(rng.integers(0, B))] = L
    lens[int(rng.integers(0, B))] = max(300, int(lens.min()) // 3)
    for b in range(B):
        sig[b, lens[b]:] = 0
    ref = O.forward_all(sig, lens, enc_sd, dec_sd, jas)
    eng.handle.profile_begin()
    r = eng.forward(torch.from_numpy(sig).cuda(), torch.from_numpy(lens).cuda(), want_logp=True)
    torch.cuda.synchronize()
    STATS["fused_launches"] += int(eng.handle.profile_end()["fused"]["launches"])
    lp, want = r["logp"].cpu(), ref["logp"]
    tol = max(5e-4, 2e-5 * float(want.abs().max()))
    err = float((lp - want).abs().max()) if lp.shape == want.shape else float("inf")
    top2 = want.topk(2, -1).values
    clear = (top2[..., 0] - top2[..., 1]) > 2 * tol
    STATS["cases"] += 1
    STATS["rows"] += B
    STATS["by_batch_class"][cls] += 1
    STATS["worst_err_over_tol"] = max(STATS["worst_err_over_tol"], err / tol)
    ok = (err <= tol and bool(torch.isfinite(r["logp"]).all()) and r["enc_len"].cpu().tolist() == ref["enc_len"].tolist()
          and bool((r["pred"].cpu()[clear] == ref["pred"][clear]).all()))
    if ok:
        return None
    return f"encoder case {case}: gemm {gemm} B {B} L {L} err {err:.3e} tol {tol:.3e} blocks {[(b['filters'], b['kernel'][0], b['repeat'], b['stride'][0], b['dilation'][0], b['residual'], b['separable']) for b in jas]}"


if __name__ == "__main__":
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    S0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    LIMIT = float(sys.argv[3]) if len(sys.argv) > 3 else 1e9
    t0, bad = time.time(), 0
    for case in range(S0, S0 + N):
        if time.time() - t0 > LIMIT:
            break
        msg = encoder_case(case)
        if msg:
            bad += 1
            print("MISMATCH", msg, flush=True)
    print(f"{STATS['cases']} cases from {S0} ({STATS['rows']} rows; batches of 1-5 / 6-20 / 21-72 rows: {STATS['by_batch_class']}; "
          f"{STATS['fused_launches']} fused launches), {bad} mismatches, worst error {STATS['worst_err_over_tol']:.2f} x the tolerance, {time.time() - t0:.0f} s")
