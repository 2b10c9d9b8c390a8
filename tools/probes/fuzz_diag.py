import sys, copy, numpy as np, torch
sys.path.insert(0, '/root/repo/tests'); sys.path.insert(0, '/root/repo')
from viet_asr_amd import configs, synth
from viet_asr_amd.engine import QuartzNetCTC
from oracle import quartznet_oracle as O
def blk(filters, kernel, repeat, stride=1, residual=False, separable=True):
    return dict(filters=filters, repeat=repeat, kernel=[kernel], stride=[stride], dilation=[1], dropout=0.0, residual=residual, separable=separable)
for seed in (1, 5, 6):
    rng = np.random.default_rng(4000 + seed)
    cfg = copy.deepcopy(configs.builtin("quartznet15x5"))
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
    eng = QuartzNetCTC(cfg, enc_sd, dec_sd, gemm="fp32")
    B, L = int(rng.integers(1, 7)), int(rng.integers(3000, 60000))
    sig, lens = synth.audio_batch(B, L, seed, ragged=True)
    lens[int(rng.integers(0, B))] = L
    lens[int(rng.integers(0, B))] = max(400, int(lens.min()) // 4)
    for b in range(B):
        sig[b, lens[b]:] = 0
    sc = float(rng.choice([1e-3, 1.0, 30.0]))
    sig[0] *= sc
    ref = O.forward_all(sig, lens, enc_sd, dec_sd, jas)
    r = eng.forward(torch.from_numpy(sig).cuda(), torch.from_numpy(lens).cuda(), want_logp=True)
    d = (r["logp"].cpu() - ref["logp"]).abs()
    print("seed", seed, "scale of row 0", sc, "lens", lens.tolist(), "enc_len", ref["enc_len"].tolist())
    for b in range(B):
        n = int(ref["enc_len"][b])
        print("   row", b, "err valid %.3e" % float(d[b, :n].max()), "err padded %.3e" % (float(d[b, n:].max()) if n < d.shape[1] else 0.0))
    # same rows with sig[0] at unit scale
    sig2 = sig.copy(); sig2[0] /= sc
    ref2 = O.forward_all(sig2, lens, enc_sd, dec_sd, jas)
    r2 = eng.forward(torch.from_numpy(sig2).cuda(), torch.from_numpy(lens).cuda(), want_logp=True)
    print("   with row 0 at unit scale: err %.3e" % float((r2["logp"].cpu() - ref2["logp"]).abs().max()))
    # mel stage
    from viet_asr_amd import _lib, stages
    from viet_asr_amd.frontend_tables import frontend_description
    h = _lib.Handle(frontend=frontend_description(cfg["AudioToMelSpectrogramPreprocessor"])); h.finalize()
    mel, seq = stages.melspec(h, torch.from_numpy(sig).cuda(), torch.from_numpy(lens).cuda())
    mref, _ = O.melspec_forward(sig, lens)
    dm = (mel.cpu() - mref).abs()
    print("   mel err per row", [float(dm[b].max()) for b in range(B)])
