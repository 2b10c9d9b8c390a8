#!/usr/bin/env python3
"""Diagnostic (dev): configs[4]-shaped ragged batch, device vs oracle on sampled rows, error split into valid / padded frames,
per GEMM arithmetic and kernel-path switch (VASR_LIB_PATH = devtools library for the switches)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import viet_asr_amd  # noqa
from viet_asr_amd import audio, configs, synth
from viet_asr_amd.engine import QuartzNetCTC
from oracle import quartznet_oracle as O
B = int(os.environ.get("B", "512")); step = int(os.environ.get("STEP", "32"))
cfg = configs.builtin("quartznet15x5"); jas = cfg["JasperEncoder"]["jasper"]
enc_sd, dec_sd = synth.encoder_state_dict(jas, 64, 5), synth.decoder_state_dict(1024, 29, 5)
sig, lens = synth.audio_batch(B, 240000, 5, ragged=True)
gpu = torch.device("cuda:0")
x16, l16 = audio.resample(torch.from_numpy(sig).to(gpu), torch.from_numpy(lens).to(gpu), 8000, 16000)
rows = torch.arange(0, B, step)
ref = O.forward_all(x16[rows.to(gpu)].cpu().numpy(), l16[rows.to(gpu)].cpu().numpy(), enc_sd, dec_sd, jas)
el = ref["enc_len"].long()
T1 = ref["logp"].shape[1]
valid = torch.arange(T1)[None, :] < el[:, None]
for gemm in os.environ.get("GEMMS", "f16x2,fp32").split(","):
    eng = QuartzNetCTC(cfg, enc_sd, dec_sd, gemm=gemm)
    r = eng.forward(x16, l16, want_logp=True)
    lp = r["logp"][rows.to(gpu)].cpu()
    d = (lp - ref["logp"]).abs().amax(-1)
    print(gemm, "valid-frame err %.4g (scale %.4g)  padded-frame err %.4g (scale %.4g)" % (
        float(d[valid].max()), float(ref["logp"][valid].abs().max()), float(d[~valid].max()) if (~valid).any() else 0.0,
        float(ref["logp"][~valid].abs().max()) if (~valid).any() else 0.0))
    worst = d.amax(1)
    k = int(worst.argmax())
    t = int(d[k].argmax())
    print("   worst row", int(rows[k]), "enc_len", int(el[k]), "frame", t, "err", float(d[k, t]), "flips", int((r["pred"][rows.to(gpu)].cpu() != ref["pred"]).sum()))
    # error profile of the worst row over frame position (in blocks of 128 frames)
    print("   per-128-frame block max err of that row:", [round(float(d[k, i:i + 128].max()), 4) for i in range(0, T1, 128)])
    del eng, r
    torch.cuda.empty_cache()
