#!/usr/bin/env python3
"""Config-4 scheduling probe (dev): the search of one batch per step on 64 CUs (shipped), against the searches of TWO
batches in one launch every other step, two utterances per CU on a CU-masked stream.   MODE=base|group|group_nomask|half"""
import ctypes, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import viet_asr_amd
from viet_asr_amd import configs, synth
from viet_asr_amd.beam import BeamSearchDecoder
from viet_asr_amd.engine import QuartzNetCTC
import tempfile
mode = os.environ.get("MODE", "base"); steps = int(os.environ.get("STEPS", 20)); ncu = int(os.environ.get("NCU", 64))
cfg = configs.builtin("quartznet15x5"); jas = cfg["JasperEncoder"]["jasper"]
eng = QuartzNetCTC(cfg, synth.encoder_state_dict(jas, 64, 3), synth.decoder_state_dict(1024, 29, 3))
lm = os.path.join(tempfile.mkdtemp(), "lm.arpa"); synth.synthetic_arpa(lm, cfg["labels"], seed=3)
dec = BeamSearchDecoder(cfg["labels"], lm_path=lm, alpha=0.5, beta=1.5)
sig, lens = synth.audio_batch(64, 160000, 3)
wav, ln = torch.from_numpy(sig).cuda(), torch.from_numpy(lens).cuda()
hip = ctypes.CDLL("libamdhip64.so")
def masked_stream(n):
    words = [0] * 8
    for i in range(n): words[i // 32] |= 1 << (i % 32)
    s = ctypes.c_void_p(); arr = (ctypes.c_uint32 * 8)(*words)
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), 8, arr); assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value)
main = torch.cuda.current_stream()
def run(n):
    if mode == "base":
        for _ in range(n): r = eng.forward_beam(wav, ln, dec, 128)
        return
    side = masked_stream(ncu) if mode != "group_nomask" else torch.cuda.Stream()
    G = 1 if mode == "half" else 2
    r0 = eng.forward(wav, ln, want_logp=True)
    big = torch.empty((G * 64,) + tuple(r0["logp"].shape[1:]), device="cuda")
    done = None
    for k in range(n):
        busy = ncu if (done is not None and not done.query()) else 0
        eng.handle.set_busy_cus(busy)
        r = eng.forward(wav, ln, want_logp=True)
        eng.handle.set_busy_cus(0)
        if done is not None and k % G == 0: main.wait_event(done)        # the halves are about to be overwritten
        big[(k % G) * 64:(k % G + 1) * 64].copy_(r["logp"])
        if k % G == G - 1:
            ev = torch.cuda.Event(); ev.record(main)
            with torch.cuda.stream(side):
                side.wait_event(ev)
                dec.decode_ids(big, 128)
                done = torch.cuda.Event(); done.record(side)
run(4); torch.cuda.synchronize()
t0 = time.perf_counter(); run(steps); torch.cuda.synchronize()
print(f"{mode} ncu={ncu}: {(time.perf_counter() - t0) / steps * 1e3:.3f} ms per step of 64 x 10 s", flush=True)
