#!/usr/bin/env python3
"""Randomised device-vs-oracle check of the mel front end (dev tool): batch shapes, ragged lengths, hop multiples,
rows shorter than the reflect padding, silent rows, loud rows."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import viet_asr_amd  # noqa
from viet_asr_amd import _lib, configs, stages
from viet_asr_amd.frontend_tables import frontend_description
from oracle import quartznet_oracle as O

cfg = configs.builtin("quartznet15x5")
pre = dict(cfg["AudioToMelSpectrogramPreprocessor"])
hp = _lib.Handle(frontend=frontend_description(pre))
hp.finalize()
hraw = _lib.Handle(frontend=frontend_description(dict(pre, normalize=None)))      # log-mel before normalisation
hraw.finalize()


def frontend_case(case, tol=2e-4):
    r = np.random.RandomState(40_000 + case)
    B = int(r.randint(1, 7))
    L = int(r.choice([r.randint(257, 2000), r.randint(2000, 40000), 160 * r.randint(2, 200), 160 * r.randint(2, 200) + 1]))
    lens = r.randint(2, L + 1, size=B).astype(np.int64)
    lens[r.randint(B)] = L
    if B > 1:
        lens[r.randint(B)] = int(r.choice([2, 3, 159, 160, 161, 320, 321]))      # seq_len 1..3: NaN / tiny-sample statistics
        lens[r.randint(B)] = max(2, (L // 160) * 160)                            # exact hop multiple (quirk Q2)
        lens[int(np.argmax(lens))] = L
    amp = float(r.choice([1e-4, 0.1, 0.9]))
    x = np.zeros((B, L), dtype=np.float32)
    for b in range(B):
        x[b, : lens[b]] = (amp * r.randn(lens[b])).astype(np.float32)
    if B > 2:
        x[1] = 0.0                                                               # a silent row: log of the guard value
    xd, ld = torch.from_numpy(x).cuda(), torch.from_numpy(lens).cuda()
    raw, seq = stages.melspec(hraw, xd, ld)
    mel, _ = stages.melspec(hp, xd, ld)
    wraw, wseq = O.melspec_forward(x, lens, normalize=None)
    want, _ = O.melspec_forward(x, lens)
    raw, mel, seq = raw.cpu(), mel.cpu(), seq.cpu()
    if not torch.equal(seq, wseq):
        return f"frontend case {case}: seq {seq.tolist()} vs {wseq.tolist()}"
    if mel.shape != want.shape or raw.shape != wraw.shape:
        return f"frontend case {case}: shape {tuple(mel.shape)} vs {tuple(want.shape)}"
    # (1) log-mel before normalisation.  A float32 FFT's rounding error is ABSOLUTE -- a few 2^-24 of the frame's amplitude --, so a
    # bin far below its frame's peak carries it as a large RELATIVE error, i.e. as an error of its logarithm: two correct float32
    # front ends differ there (the float32 oracle itself is 2.5e-4 off the float64 front end at a bin 18 nats below the peak,
    # round-6 campaign, case 100997).  Bound per element: tol + 16 * 2^-24 * (peak amplitude / bin amplitude) -- 1e-5 on top of tol for
    # bins within 6 nats of the peak, where it matters.
    d = (raw - wraw).abs()
    peak = wraw.max(dim=1, keepdim=True).values
    lim = tol + 16 * 2.0 ** -24 * torch.exp(0.5 * (peak - wraw).clamp(min=0.0).double()).float()
    valid = (torch.arange(raw.shape[-1])[None, :] < wseq[:, None])[:, None, :]
    over = (d > lim) & valid
    if over.any():
        b, f, t = [int(v) for v in torch.nonzero(over)[0]]
        return (f"frontend case {case}: raw log-mel row {b} bin {f} frame {t}: {raw[b, f, t].item():.6f} vs {wraw[b, f, t].item():.6f} "
                f"(limit {lim[b, f, t].item():.2e}, frame peak {peak[b, 0, t].item():.2f}; B {B} L {L} lens {lens.tolist()} amp {amp})")
    # (2) normalised: (x - mean) / (std + 1e-5) turns an error e of x into e / std, and rows whose log-mel barely moves
    # (digital silence: std = 0; bins under the log guard) have std << 1: the bound follows the row's own std
    nan_w, nan_m = torch.isnan(want), torch.isnan(mel)
    if not torch.equal(nan_w, nan_m):
        return f"frontend case {case}: NaN pattern differs (B {B} L {L} lens {lens.tolist()})"
    for b in range(B):
        n = int(wseq[b])
        if n < 2:
            continue
        std = wraw[b, :, :n].double().std(dim=1) + 1e-5
        e_raw = (raw[b, :, :n] - wraw[b, :, :n]).abs().max(dim=1).values.double()      # what (1) let through for this bin
        # the reference's OWN mean is a float32 sum (features.py:21-23 -> ATen's vectorised cascade sum): on a row of 216 equal values
        # -- digital silence: every frame log(2^-24) -- it is 3 ulp = 5.7e-6 off the value itself, which (x - mean) / (0 + 1e-5)
        # turns into a constant -0.572 where the exact answer (and the device's, whose statistics are in double) is 0
        e_mean = (wraw[b, :, :n].mean(dim=1).double() - wraw[b, :, :n].double().mean(dim=1)).abs()
        bound = (tol + (2 * e_raw + 2 * e_mean + 4e-6) / std).float()[:, None]
        bad = ((mel[b, :, :n] - want[b, :, :n]).abs() > bound)
        if bad.any():
            f, t = [int(v[0]) for v in torch.nonzero(bad)[0:1].T]
            return (f"frontend case {case}: row {b} bin {f} frame {t}: {mel[b, f, t].item():.6f} vs {want[b, f, t].item():.6f}, "
                    f"row std {std[f].item():.2e} (B {B} L {L} lens {lens.tolist()} amp {amp})")
    masked = torch.arange(mel.shape[-1])[None, :] >= wseq[:, None]
    if (mel.transpose(1, 2)[masked] != 0).any():
        return f"frontend case {case}: masked frames not exactly zero"
    return None


if __name__ == "__main__":
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    S0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    LIMIT = float(sys.argv[3]) if len(sys.argv) > 3 else 1e9      # seconds: stop early, still print the summary
    t0, bad, done = time.time(), 0, 0
    for case in range(S0, S0 + N):
        if time.time() - t0 > LIMIT:
            break
        msg = frontend_case(case)
        done += 1
        if msg:
            bad += 1
            print("MISMATCH", msg, flush=True)
    print(f"{done} cases from {S0}, {bad} mismatches, {time.time() - t0:.0f} s")
