#!/usr/bin/env python3
"""Small-batch latency with and without hipGraph replay of the fused path (dev tool)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import viet_asr_amd
from viet_asr_amd import configs, synth
from viet_asr_amd.engine import QuartzNetCTC
cfg = configs.builtin("quartznet15x5"); jas = cfg["JasperEncoder"]["jasper"]
eng = QuartzNetCTC(cfg, synth.encoder_state_dict(jas, 64, 3), synth.decoder_state_dict(1024, 29, 3))
for B in (1, 4, 16):
    sig, lens = synth.audio_batch(B, 160000, 3)
    wav, ln = torch.from_numpy(sig).cuda(), torch.from_numpy(lens).cuda()
    for _ in range(3): r = eng.forward(wav, ln)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50): r = eng.forward(wav, ln)
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / 50
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(2): eng.forward(wav, ln)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        rg = eng.forward(wav, ln)
    g.replay(); torch.cuda.synchronize()
    same = torch.equal(rg["ids"], r["ids"]) and torch.equal(rg["id_len"], r["id_len"])
    t0 = time.perf_counter()
    for _ in range(50): g.replay()
    torch.cuda.synchronize()
    graph = (time.perf_counter() - t0) / 50
    print(f"B={B}: eager {eager*1e3:.3f} ms, hipGraph replay {graph*1e3:.3f} ms, identical results: {same}")
