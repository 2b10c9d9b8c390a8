
import sys, copy, numpy as np, torch
sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo')
from viet_asr_amd import configs, synth
from viet_asr_amd.engine import QuartzNetCTC
from oracle import quartznet_oracle as O
bad = []
def blk(filters, kernel, repeat, stride=1, residual=False, separable=True):
    return dict(filters=filters, repeat=repeat, kernel=[kernel], stride=[stride], dilation=[1], dropout=0.0,
                residual=residual, separable=separable)
for seed in range(10):
    rng = np.random.default_rng(4000 + seed)
    cfg = copy.deepcopy(configs.builtin("quartznet15x5"))
    # 256-channel blocks with the two kernel widths the fused kernel covers, repeats 1-5 (a lone sub-block is a residual
    # sub-block), with and without residual, behind a strided or unstrided prologue; a 512-channel block in between
    jas = [blk(256, 33, 1, stride=int(rng.choice([1, 2])))]
    for _ in range(int(rng.integers(1, 4))):
        jas.append(blk(256, int(rng.choice([33, 39])), int(rng.integers(1, 6)), residual=bool(rng.random() < 0.7)))
        if rng.random() < 0.3:
            jas.append(blk(512, 51, 1, residual=True))
            jas.append(blk(256, 39, int(rng.integers(1, 3)), residual=True))
    jas.append(blk(int(rng.choice([128, 256])), 1, 1, separable=False))
    cfg["JasperEncoder"]["jasper"] = jas
    enc_sd = synth.encoder_state_dict(jas, 64, seed)
    dec_sd = synth.decoder_state_dict(jas[-1]["filters"], len(cfg["labels"]) + 1, seed)
    eng = QuartzNetCTC(cfg, enc_sd, dec_sd)
    B, L = int(rng.integers(1, 7)), int(rng.integers(3000, 60000))
    sig, lens = synth.audio_batch(B, L, seed, ragged=True)
    lens[int(rng.integers(0, B))] = L
    lens[int(rng.integers(0, B))] = max(400, int(lens.min()) // 4)      # rows with whole tiles past their length
    for b in range(B):
        sig[b, lens[b]:] = 0
    sig[0] *= float(rng.choice([1e-3, 1.0, 30.0]))                       # per-utterance scales of the bound-based split
    ref = O.forward_all(sig, lens, enc_sd, dec_sd, jas)
    eng.handle.profile_begin()
    r = eng.forward(torch.from_numpy(sig).cuda(), torch.from_numpy(lens).cuda(), want_logp=True)
    torch.cuda.synchronize()
    fused = eng.handle.profile_end()["fused"]["launches"]
    want = ref["logp"]
    tol = max(5e-4, 2e-5 * float(want.abs().max()))
    err = float((r["logp"].cpu() - want).abs().max())
    top2 = want.topk(2, -1).values
    clear = (top2[..., 0] - top2[..., 1]) > 2 * tol
    ok = err <= tol and bool((r["pred"].cpu()[clear] == ref["pred"][clear]).all()) and \
        r["enc_len"].cpu().tolist() == ref["enc_len"].tolist() and bool(torch.isfinite(r["logp"]).all())
    want_fused = sum(b["repeat"] for b in jas[1:] if b["separable"] and b["filters"] == 256)
    # (the first sub-block after a 512-channel block has 512 input channels: not a fused shape)
    print(seed, "err", err, "tol", tol, "fused", fused, "scale", float(want.abs().max()), "ok", ok)
    if not ok:
        bad.append((seed, err, tol, fused, want_fused, B, L))
print("FUSED_FUZZ_OK" if not bad else "FUSED_FUZZ_BAD %r" % bad)
