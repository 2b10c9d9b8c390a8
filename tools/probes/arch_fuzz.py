#!/usr/bin/env python3
"""Wide random-architecture / shape fuzz against the oracle (dev tool; the suite keeps 16 + 8 of these cases):
    python tools/probes/arch_fuzz.py [first_seed] [count]      GEMM=f16x2|bf16x3|fp32, VASR_FUSED_MIN_TILES=1 forces the fused kernel
Architectures mix what the library accepts: 128-512 channels, odd kernels 3-99, repeats 1-4, dilation, stride-2 prologue,
residual blocks, 256-channel K = 33 / 39 blocks (the fused shapes) next to wider ones; batches 1-24, ragged lengths, one row
far shorter than the rest, one row at another level."""
import copy, os, sys
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from viet_asr_amd import configs, synth
from viet_asr_amd.engine import QuartzNetCTC
from oracle import quartznet_oracle as O
first, count = (int(sys.argv[1]) if len(sys.argv) > 1 else 0), (int(sys.argv[2]) if len(sys.argv) > 2 else 40)
gemm = os.environ.get("GEMM", "f16x2")
def blk(filters, kernel, repeat, stride=1, dilation=1, residual=False, separable=True):
    return dict(filters=int(filters), repeat=int(repeat), kernel=[int(kernel)], stride=[int(stride)], dilation=[int(dilation)],
                dropout=0.0, residual=bool(residual), separable=bool(separable))
bad = []
for seed in range(first, first + count):
    rng = np.random.default_rng(77000 + seed)
    odd = lambda lo, hi: int(rng.integers(lo // 2, hi // 2 + 1)) * 2 + 1
    jas = [blk(rng.choice([128, 256]), odd(3, 41), rng.integers(1, 3), stride=rng.choice([1, 2]))]
    for _ in range(int(rng.integers(1, 5))):
        if rng.random() < 0.4:
            jas.append(blk(256, rng.choice([33, 39]), rng.integers(1, 5), residual=rng.random() < 0.6))
        else:
            jas.append(blk(rng.choice([128, 256, 384, 512]), rng.choice([odd(3, 99), 51, 63, 75]), rng.integers(1, 4), residual=rng.random() < 0.6))
    if rng.random() < 0.6:
        jas.append(blk(rng.choice([256, 512]), rng.choice([odd(3, 91), 87]), 1, dilation=2))
    jas.append(blk(rng.choice([128, 384, 1024]), 1, 1, separable=False))
    cfg = copy.deepcopy(configs.builtin("quartznet15x5"))
    cfg["JasperEncoder"]["jasper"] = jas
    enc_sd = synth.encoder_state_dict(jas, 64, seed)
    dec_sd = synth.decoder_state_dict(jas[-1]["filters"], len(cfg["labels"]) + 1, seed)
    eng = QuartzNetCTC(cfg, enc_sd, dec_sd, gemm=gemm)
    B, L = int(rng.choice([1, 2, 3, 5, 8, 13, 24])), int(rng.integers(1500, 50000))
    sig, lens = synth.audio_batch(B, L, seed, ragged=True)
    lens[int(rng.integers(0, B))] = L
    lens[int(rng.integers(0, B))] = max(300, int(lens.min()) // 4)
    for b in range(B):
        sig[b, lens[b]:] = 0
    sig[int(rng.integers(0, B))] *= float(rng.choice([0.03, 1.0, 30.0]))
    ref = O.forward_all(sig, lens, enc_sd, dec_sd, jas)
    r = eng.forward(torch.from_numpy(sig).cuda(), torch.from_numpy(lens).cuda(), want_logp=True)
    want = ref["logp"]
    tol = max(5e-4, 2e-5 * float(want.abs().max()))
    err = float((r["logp"].cpu() - want).abs().max())
    top2 = want.topk(2, -1).values
    clear = (top2[..., 0] - top2[..., 1]) > 2 * tol
    ok = err <= tol and bool((r["pred"].cpu()[clear] == ref["pred"][clear]).all()) and \
        r["enc_len"].cpu().tolist() == ref["enc_len"].tolist() and bool(torch.isfinite(r["logp"]).all())
    print(f"seed {seed} B={B} L={L} blocks={[(b['filters'], b['kernel'][0], b['repeat'], b['residual']) for b in jas]} err {err:.2e} tol {tol:.2e} ok {ok}", flush=True)
    if not ok:
        bad.append(seed)
    del eng
print("ARCH_FUZZ_OK" if not bad else f"ARCH_FUZZ_BAD {bad}")
