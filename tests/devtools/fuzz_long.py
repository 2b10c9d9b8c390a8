#!/usr/bin/env python3
"""engine.forward_long (one recording in bounded memory: halo windows) against the one-pass call (dev tool, GPU; no oracle):
python tests/devtools/fuzz_long.py [n] [seed0] [max_seconds]

Random model (the shipped 12x1 or a random block list), random length 0.3-60 s, random window size (16-3000 output frames) and
rows per pass (1-6).  In the fp32 and 3 x bf16 arithmetics every kept frame sees exactly the one-pass inputs: log-probs, predictions,
ids BIT FOR BIT.  In the 2 x fp16 arithmetic the operand scale follows the row a kernel works on (a window / the recording): log-probs
within the parity tolerance, predictions equal wherever the one-pass margin exceeds twice the tolerance."""
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

STATS = {"cases": 0, "exact_cases": 0, "windows": 0, "f16x2_pred_diffs_inside_margin": 0}
_ENG = {}


def long_case(case):
    rng = np.random.default_rng(500000 + case)
    gemm = str(rng.choice(["bf16x3", "fp32", "f16x2"]))
    if rng.random() < 0.5:
        if gemm not in _ENG:
            cfg = configs.builtin("quartznet12x1_vi")
            jas = cfg["JasperEncoder"]["jasper"]
            _ENG[gemm] = QuartzNetCTC(cfg, synth.encoder_state_dict(jas, 64, 4), synth.decoder_state_dict(1024, len(cfg["labels"]) + 1, 4), gemm=gemm)
        eng = _ENG[gemm]
    else:
        cfg = copy.deepcopy(configs.builtin("quartznet15x5"))
        jas = cfg["JasperEncoder"]["jasper"] = _random_architecture(rng)
        for b in jas:
            if b["stride"][0] > 1:
                b["repeat"] = 1          # (forward_long refuses a strided block with repeat > 1: halo_mel_frames)
        eng = QuartzNetCTC(cfg, synth.encoder_state_dict(jas, 64, case), synth.decoder_state_dict(jas[-1]["filters"], len(cfg["labels"]) + 1, case), gemm=gemm)
    n = int(rng.choice([int(rng.integers(4800, 48000)), int(rng.integers(48000, 320000)), int(rng.integers(320000, 960000)), 160 * int(rng.integers(100, 3000))]))
    x = (0.1 * np.random.default_rng(case).standard_normal(n)).astype(np.float32)
    cut = int(rng.integers(0, n))
    x[:cut] *= float(rng.choice([1.0, 0.05, 8.0]))                         # a level change along the recording
    xd = torch.from_numpy(x).cuda()
    chunk = int(rng.choice([int(rng.integers(16, 200)), int(rng.integers(200, 3000))]))
    rows = int(rng.integers(1, 7))
    one = eng.forward(xd[None], torch.tensor([n], device="cuda"), want_logp=True)
    r = eng.forward_long(xd, chunk_frames=chunk, rows_per_pass=rows, want_logp=True)
    STATS["cases"] += 1
    STATS["windows"] += -(-one["logp"].shape[1] // chunk)
    tag = f"long case {case}: gemm {gemm} n {n} chunk {chunk} rows {rows} frames {one['logp'].shape[1]}"
    if r["logp"].shape != one["logp"].shape or float(r["enc_len"][0]) != float(one["enc_len"][0]):
        return f"{tag}: shape {tuple(r['logp'].shape)} vs {tuple(one['logp'].shape)} / enc_len {float(r['enc_len'][0])} vs {float(one['enc_len'][0])}"
    if gemm != "f16x2":
        STATS["exact_cases"] += 1
        if not torch.equal(r["logp"], one["logp"]):
            d = (r["logp"] - one["logp"]).abs()
            t = int(torch.nonzero(d.amax(-1)[0] > 0)[0])
            return f"{tag}: log-probs differ (max {float(d.max()):.3e}, first frame {t})"
        if not torch.equal(r["pred"], one["pred"]) or int(r["id_len"][0]) != int(one["id_len"][0]):
            return f"{tag}: predictions differ"
        return None
    lp = one["logp"]
    tol = max(5e-4, 2e-5 * float(lp.abs().max()))
    err = float((r["logp"] - lp).abs().max())
    if not err <= tol:
        return f"{tag}: log-probs off by {err:.3e} (tolerance {tol:.3e})"
    top2 = lp.topk(2, -1).values
    clear = (top2[..., 0] - top2[..., 1]) > 2 * tol
    if not bool((r["pred"][clear] == one["pred"][clear]).all()):
        return f"{tag}: a prediction differs on a frame whose margin exceeds twice the tolerance"
    STATS["f16x2_pred_diffs_inside_margin"] += int((r["pred"] != one["pred"]).sum())
    return None


if __name__ == "__main__":
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    S0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    LIMIT = float(sys.argv[3]) if len(sys.argv) > 3 else 1e9
    t0, bad = time.time(), 0
    for case in range(S0, S0 + N):
        if time.time() - t0 > LIMIT:
            break
        msg = long_case(case)
        if msg:
            bad += 1
            if bad <= 12:
                print("MISMATCH", msg, flush=True)
    print(f"{STATS['cases']} cases from {S0} ({STATS['exact_cases']} in an exact arithmetic, {STATS['windows']} windows; f16x2: "
          f"{STATS['f16x2_pred_diffs_inside_margin']} frames decoded differently inside the margin), {bad} mismatches, {time.time() - t0:.0f} s")
