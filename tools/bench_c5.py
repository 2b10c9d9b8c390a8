#!/usr/bin/env python3
"""BASELINE config 5 shape on one GPU (dev tool): 8 kHz clips of 30 s -> device resample to 16 kHz -> greedy CTC."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import viet_asr_amd
from viet_asr_amd import audio, configs, synth
from viet_asr_amd.engine import QuartzNetCTC
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
cfg = configs.builtin("quartznet15x5"); jas = cfg["JasperEncoder"]["jasper"]
eng = QuartzNetCTC(cfg, synth.encoder_state_dict(jas, 64, 3), synth.decoder_state_dict(1024, 29, 3))
sig, lens = synth.audio_batch(B, 240000, 5)            # 30 s at 8 kHz
x8 = torch.from_numpy(sig).cuda(); l8 = torch.from_numpy(lens).cuda()
def run():
    x16, l16 = audio.resample(x8, l8, 8000, 16000)
    return x16, l16, eng.forward(x16, l16)
for _ in range(2): run()
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
ev[0].record(); x16, l16 = audio.resample(x8, l8, 8000, 16000); ev[1].record(); r = eng.forward(x16, l16); ev[2].record()
torch.cuda.synchronize()
t_rs, t_fw = ev[0].elapsed_time(ev[1]), ev[1].elapsed_time(ev[2])
print(f"B={B} x 30 s: resample {t_rs:.2f} ms, forward {t_fw:.2f} ms -> {B * 30 / ((t_rs + t_fw) * 1e-3):,.0f}x real time "
      f"(forward alone {B * 30 / (t_fw * 1e-3):,.0f}x); output {tuple(x16.shape)}")
