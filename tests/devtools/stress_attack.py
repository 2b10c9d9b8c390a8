#!/usr/bin/env python3
"""Every entry point next to FOREIGN 16-bit matrix kernels on another stream (dev tool, GPU):
python tests/devtools/stress_attack.py [seconds per case]

Round 6 found that a kernel of this library can return WRONG VALUES when a kernel of somebody else's runs beside it on the same
compute unit -- not through memory (no out-of-bounds access on either side; LDS and register canaries stay clean) but inside the
SIMD: packed-FP32 vector instructions (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32) of one wavefront gave wrong results while a
wavefront of a 16-bit MFMA kernel (torch's own bf16 matmul / fp16 bmm, or this library's split GEMM launched from a second handle)
shared its SIMD; the same kernel built without packed-FP32 instructions is immune (profiles/r06_concurrency.txt).  This tool is the
regression harness: thread B runs `torch.bmm` on fp16 [512, 64, 64] operands back to back (the most reliable trigger found), thread
A runs one entry point of the library over and over on fixed inputs, on a stream of its own, and compares every result with what
the same call returned when the device was otherwise idle.  Prints one line per case; any "wrong" > 0 is a defect."""
import os
import sys
import tempfile
import threading
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import viet_asr_amd  # noqa: E402,F401
from viet_asr_amd import audio, configs, stages, synth  # noqa: E402
from viet_asr_amd.beam import BeamSearchDecoder  # noqa: E402
from viet_asr_amd.engine import QuartzNetCTC  # noqa: E402


def same(a, b):
    if isinstance(a, (tuple, list)) and len(a) == 3 and a[0].dtype == torch.int32 and a[1].dtype == torch.int32 and a[0].dim() == 2:
        ids, n, score = a                        # beam search: (ids, id_len, score); ids past id_len are unwritten
        return bool(torch.equal(n, b[1]) and torch.equal(score, b[2]) and all(torch.equal(ids[r, : int(n[r])], b[0][r, : int(n[r])]) for r in range(ids.shape[0])))
    if isinstance(a, (tuple, list)):
        return all(same(x, y) for x, y in zip(a, b))
    if isinstance(a, dict):
        return all(same(a[k], b[k]) for k in a if isinstance(a[k], torch.Tensor) and k != "ids") and (
            "ids" not in a or all(torch.equal(a["ids"][r, : int(a["id_len"][r])], b["ids"][r, : int(b["id_len"][r])]) for r in range(a["ids"].shape[0])))
    if a.dtype.is_floating_point:
        return bool(torch.equal(torch.nan_to_num(a, nan=-7.0), torch.nan_to_num(b, nan=-7.0)))
    return bool(torch.equal(a, b))


def cases():
    dev = "cuda"
    out = []
    for model, shapes in (("quartznet12x1_vi", [(1, 52000), (6, 30000), (40, 9000)]), ("quartznet15x5", [(3, 40000), (64, 16000)])):
        cfg = configs.builtin(model)
        jas = cfg["JasperEncoder"]["jasper"]
        sd = synth.encoder_state_dict(jas, 64, 5), synth.decoder_state_dict(1024, len(cfg["labels"]) + 1, 5)
        for gemm in ("f16x2", "bf16x3", "fp32"):
            eng = QuartzNetCTC(cfg, sd[0], sd[1], gemm=gemm)
            for B, L in shapes:
                if gemm == "fp32" and B * L > 400000:
                    continue
                sig, lens = synth.audio_batch(B, L, 50 + B, ragged=True)
                w, n = torch.from_numpy(sig).to(dev), torch.from_numpy(lens).to(dev)
                out.append((f"{model} {gemm} forward {B} x {L}", lambda e=eng, w=w, n=n: e.forward(w, n, want_logp=True)))
                if gemm == "f16x2":
                    out.append((f"{model} {gemm} forward {B} x {L} row-independent", lambda e=eng, w=w, n=n: e.forward(w, n, want_logp=True, row_independent=True)))
        if model == "quartznet12x1_vi":
            eng = QuartzNetCTC(cfg, sd[0], sd[1])
            arpa = os.path.join(tempfile.mkdtemp(prefix="vasr_attack_"), "lm.arpa")
            synth.synthetic_arpa(arpa, cfg["labels"], n_words=2000, n_bigrams=6000, n_trigrams=6000, seed=1)
            dec = BeamSearchDecoder(cfg["labels"], lm_path=arpa, alpha=0.5, beta=1.5)
            for B, L in ((1, 60000), (20, 24000), (70, 9000)):
                sig, lens = synth.audio_batch(B, L, 70 + B, ragged=True)
                lp = eng.forward(torch.from_numpy(sig).to(dev), torch.from_numpy(lens).to(dev), want_logp=True)["logp"].clone()
                out.append((f"beam search width 64 + LM, {B} rows x {lp.shape[1]} frames", lambda d=dec, lp=lp: d.decode_ids(lp, 64)))
            h = eng.handle
            sig, lens = synth.audio_batch(8, 48000, 3, ragged=True)
            w, n = torch.from_numpy(sig).to(dev), torch.from_numpy(lens).to(dev)
            mel, seq = stages.melspec(h, w, n)
            enc, _ = stages.encoder(h, mel, seq, 1024)
            mel, seq, enc = mel.clone(), seq.clone(), enc.clone()
            out.append(("module entry: melspec 8 x 48000", lambda: stages.melspec(h, w, n)))
            out.append(("module entry: encoder", lambda: stages.encoder(h, mel, seq, 1024)))
            out.append(("module entry: decoder", lambda: stages.decoder(h, enc)))
            x8 = torch.from_numpy(synth.audio_batch(16, 80000, 9, ragged=True)[0]).to(dev)
            n8 = torch.full((16,), 80000, dtype=torch.int64, device=dev)
            out.append(("resample 8 -> 16 kHz 16 x 80000", lambda: audio.resample(x8, n8, 8000, 16000)))
            out.append(("resample 11025 -> 16000 Hz", lambda: audio.resample(x8, n8, 11025, 16000)))
            pcm = (x8 * 20000).to(torch.int16)
            out.append(("pcm16 -> float", lambda: audio.pcm16_to_float(pcm)))
    return out


def attack(fn, secs, attacker=None):
    """Runs fn() over and over on a stream of its own for `secs` seconds while another host thread keeps `attacker` (default: torch.bmm
    on fp16 [512, 64, 64]) running on a second stream -> (calls, calls whose result differs from fn() on the idle device)."""
    if attacker is None:
        ab = torch.randn(512, 64, 64, device="cuda", dtype=torch.float16)
        attacker = lambda: torch.bmm(ab, ab)     # noqa: E731
    want = fn()
    torch.cuda.synchronize()
    stop, bad, calls = [False], [0], [0]

    def a():
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            while not stop[0]:
                r = fn()
                st.synchronize()
                calls[0] += 1
                if not same(r, want):
                    bad[0] += 1

    def b():
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            while not stop[0]:
                attacker()
                st.synchronize()

    ta, tb = threading.Thread(target=a), threading.Thread(target=b)
    ta.start(); tb.start()
    time.sleep(secs)
    stop[0] = True
    ta.join(); tb.join()
    return calls[0], bad[0]


def attackers():
    dev = "cuda"
    ab = torch.randn(512, 64, 64, device=dev, dtype=torch.float16)
    a16, b16 = torch.randn(5120, 256, device=dev, dtype=torch.bfloat16), torch.randn(256, 256, device=dev, dtype=torch.bfloat16)
    a8, b8 = torch.randn(2048, 1024, device=dev, dtype=torch.float16), torch.randn(1024, 128, device=dev, dtype=torch.float16)
    out = {"bmm16": lambda: torch.bmm(ab, ab), "matmul_bf16": lambda: a16 @ b16, "matmul_f16_thin": lambda: a8 @ b8}
    # "mfma": tools/probes/mfma_attacker.hip -- nothing but v_mfma_f32_32x32x16_f16 in a loop, 128-thread workgroups with 24 KB (or
    # "mfma0": no) LDS; against the front end of commit 25e1455 the most reliable attacker of all (6 210 of 6 213 calls wrong).
    # Compiled on first use (hipcc is on the GPU box) into the system's temporary directory.
    def synthetic(lds):
        import ctypes as C, subprocess, tempfile
        so = os.path.join(tempfile.gettempdir(), "vasr_mfma_attacker.so")
        src = os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tools", "probes", "mfma_attacker.hip")
        if not os.path.exists(so):
            subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-shared", "-fPIC", src, "-o", so], check=True, capture_output=True)
        lib = C.CDLL(so)
        lib.mfma_attacker_launch.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        sink = torch.zeros(1 << 18, device=dev)
        return lambda: lib.mfma_attacker_launch(2048, lds, 600, 0, sink.data_ptr(), None, torch.cuda.current_stream().cuda_stream)
    out["mfma"] = lambda f=[None]: (f.__setitem__(0, f[0] or synthetic(24576)), f[0]())[1]
    out["mfma0"] = lambda f=[None]: (f.__setitem__(0, f[0] or synthetic(0)), f[0]())[1]
    return out


def main():
    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 4.0
    which = sys.argv[2] if len(sys.argv) > 2 else "bmm16"      # bmm16 | matmul_bf16 | matmul_f16_thin | mfma | mfma0
    att = attackers()[which]
    print(f"attacker: {which}")
    total_bad = 0
    for name, fn in cases():
        calls, bad = attack(fn, secs, att)
        total_bad += bad
        print(f"{name:70s}: calls {calls:6d} wrong {bad}", flush=True)
    print(f"TOTAL wrong {total_bad}")


if __name__ == "__main__":
    main()
