#!/usr/bin/env python3
"""Diagnostic (dev): configs[4]-shaped ragged batch -- which stage differs from the oracle on the short rows?"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import viet_asr_amd  # noqa
from viet_asr_amd import audio, configs, stages, synth
from viet_asr_amd.engine import QuartzNetCTC
from oracle import quartznet_oracle as O
B = int(os.environ.get("B", "512"))
cfg = configs.builtin("quartznet15x5"); jas = cfg["JasperEncoder"]["jasper"]
enc_sd, dec_sd = synth.encoder_state_dict(jas, 64, 5), synth.decoder_state_dict(1024, 29, 5)
sig, lens = synth.audio_batch(B, 240000, 5, ragged=True)
gpu = torch.device("cuda:0")
x16, l16 = audio.resample(torch.from_numpy(sig).to(gpu), torch.from_numpy(lens).to(gpu), 8000, 16000)
rows = [int(r) for r in os.environ.get("ROWS", "160,0,32,64,96,128").split(",")]
ridx = torch.tensor(rows, device=gpu)
xs, ls = x16[ridx].cpu().numpy(), l16[ridx].cpu().numpy()
print("lens", ls.tolist(), "tail nonzero beyond len:", [int(np.count_nonzero(xs[k, ls[k]:])) for k in range(len(rows))])
eng = QuartzNetCTC(cfg, enc_sd, dec_sd)
h = eng.handle
mel_d, seq_d = stages.melspec(h, x16, l16)
mel_o, seq_o = O.melspec_forward(xs, ls)
print("seq equal", (seq_d[ridx].cpu() == seq_o).all().item())
for k, r in enumerate(rows):
    d = (mel_d[r].cpu() - mel_o[k]).abs()
    print("row", r, "mel err", float(d.max()), "at frame", int(d.amax(0).argmax()), "of seq", int(seq_o[k]), " mean|d|", float(d.mean()))
# the same rows as a batch of their own through the device front end
mel_s, _ = stages.melspec(h, x16[ridx].contiguous(), l16[ridx].contiguous())
print("device mel: in-batch vs sub-batch max diff", float((mel_s - mel_d[ridx]).abs().max()))
# whole path: in-batch vs sub-batch
r_all = eng.forward(x16, l16, want_logp=True)
r_sub = eng.forward(x16[ridx].contiguous(), l16[ridx].contiguous(), want_logp=True)
print("device logp: in-batch vs sub-batch max diff per row", (r_all["logp"][ridx] - r_sub["logp"]).abs().amax((1, 2)).tolist())
ref = O.forward_all(xs, ls, enc_sd, dec_sd, jas)
print("sub-batch device vs oracle per row", (r_sub["logp"].cpu() - ref["logp"]).abs().amax((1, 2)).tolist())
print("in-batch device vs oracle per row", (r_all["logp"][ridx].cpu() - ref["logp"]).abs().amax((1, 2)).tolist())
# oracle on the device's mel: does the encoder agree when fed the same features?
e_o, _ = O.encoder_forward(mel_d[ridx].cpu(), seq_o, enc_sd, jas)
lp_o = O.decoder_forward(e_o, dec_sd)
print("oracle(encoder+head) on DEVICE mel vs device logp per row", (r_all["logp"][ridx].cpu() - lp_o).abs().amax((1, 2)).tolist())
