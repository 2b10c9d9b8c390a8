#!/usr/bin/env python3
"""Does the beam search pack two utterances per compute unit?  Time B = 64 / 256 / 512 rows of CTC-like posteriors (dev)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import viet_asr_amd
from viet_asr_amd import configs, synth
from viet_asr_amd.beam import BeamSearchDecoder
cfg = configs.builtin("quartznet15x5")
words = ["xin", "chao", "viet", "nam", "toi", "la", "mot", "hai", "ba", "bon"]
lp64 = torch.from_numpy(synth.ctc_like_log_probs(64, 501, cfg["labels"], words, seed=4)).cuda()
dec = BeamSearchDecoder(cfg["labels"])
for B in (64, 128, 256, 512):
    lp = lp64.repeat(B // 64, 1, 1).contiguous()
    dec.decode_ids(lp, 128); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): dec.decode_ids(lp, 128)
    torch.cuda.synchronize()
    print(f"B={B}: {(time.perf_counter() - t0) / 3 * 1e3:.2f} ms", flush=True)
