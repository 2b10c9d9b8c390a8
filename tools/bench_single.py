#!/usr/bin/env python3
"""Latency of one utterance, the reference's own usage (infer.py: VietASR.transcribe, batch 1) (dev tool):
NeuralModule DAG path (greedy and beam wiring) against the fused one-call path."""
import os, sys, tempfile, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import viet_asr_amd  # noqa
from viet_asr_amd import configs, synth
from viet_asr_amd.infer import VietASR

model = sys.argv[1] if len(sys.argv) > 1 else "quartznet15x5"
cfg = configs.builtin(model)
jas = cfg["JasperEncoder"]["jasper"]
tmp = tempfile.mkdtemp()
enc_p, dec_p = os.path.join(tmp, "JasperEncoder-STEP-1.pt"), os.path.join(tmp, "JasperDecoderForCTC-STEP-1.pt")
torch.save({k: torch.as_tensor(v) for k, v in synth.encoder_state_dict(jas, 64, 3).items()}, enc_p)
torch.save({k: torch.as_tensor(v) for k, v in synth.decoder_state_dict(1024, len(cfg["labels"]) + 1, 3).items()}, dec_p)
x = synth.audio_batch(1, 160000, 3)[0][0]


def timeit(fn, n=20):
    fn(); fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


greedy = VietASR(model, enc_p, dec_p, device="gpu", decoder="greedy")
beam = VietASR(model, enc_p, dec_p, device="gpu", decoder="beam", beam_width=50, lm_path=None)
print(f"{model}, one 10 s utterance, host array in -> str out")
print(f"  transcribe() greedy, NeuralModule DAG : {timeit(lambda: greedy.transcribe(x)):6.2f} ms")
print(f"  transcribe_batch([x]) fused           : {timeit(lambda: greedy.transcribe_batch([x])):6.2f} ms")
print(f"  transcribe() beam 50, NeuralModule DAG: {timeit(lambda: beam.transcribe(x)):6.2f} ms")
print(f"  transcribe_batch([x], decoder='beam') : {timeit(lambda: beam.transcribe_batch([x], decoder='beam', row_independent=True)):6.2f} ms")
