#!/usr/bin/env python3
"""fp32-MFMA vs 3xbf16 GEMM: isolated layer timing + error against an fp64 reference (dev tool)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import viet_asr_amd
from viet_asr_amd import _lib, configs, synth
from viet_asr_amd.engine import QuartzNetCTC
from oracle import quartznet_oracle as O
dev = torch.device("cuda:0")
for model, B, L, seed in (("quartznet12x1_vi", 3, 32480, 1), ("quartznet15x5", 2, 20321, 3), ("quartznet15x5", 4, 160000, 7)):
    cfg = configs.builtin(model); jas = cfg["JasperEncoder"]["jasper"]
    enc_sd = synth.encoder_state_dict(jas, 64, seed); dec_sd = synth.decoder_state_dict(1024, len(cfg["labels"]) + 1, seed)
    sig, lens = synth.audio_batch(B, L, seed, True)
    ref = O.forward_all(sig, lens, enc_sd, dec_sd, jas)
    eng = QuartzNetCTC(cfg, enc_sd, dec_sd)
    wav, ln = torch.from_numpy(sig).to(dev), torch.from_numpy(lens).to(dev)
    for mode in ("fp32", "bf16x3"):
        eng.handle.set_gemm_mode(mode)
        r = eng.forward(wav, ln, want_logp=True); torch.cuda.synchronize()
        err = (r["logp"].cpu() - ref["logp"]).abs().max().item()
        mism = (r["pred"].cpu() != ref["pred"]).sum().item()
        print(f"{model} B={B} L={L} {mode:7s}: logp err vs oracle {err:.3e}  pred mismatches {mism}/{ref['pred'].numel()}  scale {ref['logp'].abs().max():.1f}")
