#!/usr/bin/env python3
"""VietASR.transcribe_manifest against one-file-at-a-time transcribe (dev tool, GPU): python tests/devtools/fuzz_manifest.py [rounds] [seed]

transcribe_manifest sorts a NeMo manifest by duration, cuts it into batches, pushes them through the pipelined engine two at a time
(row-independent) and promises, per entry, "what transcribe returns for that file alone".  Here: random manifests of 20-90 PCM WAV
files (16 / 8 / 11.025 kHz, 8 / 16 / 24 / 32-bit, mono or stereo, 0.3-9 s), random batch sizes, every transcript against
asr.transcribe(read_wav(file)) -- the reference CLI's loop (infer.py:194-206)."""
import json
import os
import sys
import tempfile
import time
import wave

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import viet_asr_amd  # noqa: E402,F401
from viet_asr_amd import audio  # noqa: E402
import fuzz_dag  # noqa: E402


def write_pcm(path, x, sr, width, channels):
    x = np.clip(x, -0.999, 0.999)
    if channels == 2:
        x = np.stack([x, 0.5 * x[::-1].copy()], axis=1).reshape(-1)
    if width == 1:
        raw = (np.round(x * 127) + 128).astype(np.uint8).tobytes()
    elif width == 2:
        raw = np.round(x * 32767).astype("<i2").tobytes()
    elif width == 3:
        v = np.round(x * (2 ** 23 - 1)).astype(np.int32)
        raw = np.stack([(v & 255), (v >> 8) & 255, (v >> 16) & 255], axis=1).astype(np.uint8).tobytes()
    else:
        raw = np.round(x * (2 ** 31 - 1)).astype("<i4").tobytes()
    with wave.open(path, "wb") as w:
        w.setnchannels(channels); w.setsampwidth(width); w.setframerate(sr); w.writeframes(raw)


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    asr = fuzz_dag.asr_for("quartznet12x1_vi")
    rng = np.random.default_rng(seed)
    t0, files, bad = time.time(), 0, []
    for rnd in range(rounds):
        d = tempfile.mkdtemp(prefix="vasr_manifest_")
        man = os.path.join(d, "manifest.json")
        entries = []
        for i in range(int(rng.integers(20, 90))):
            sr = int(rng.choice([16000, 16000, 8000, 11025]))
            n = int(rng.integers(int(0.3 * sr), 9 * sr))
            x = float(rng.choice([0.01, 0.1, 0.6])) * rng.standard_normal(n)
            p = os.path.join(d, f"f{i}.wav")
            write_pcm(p, x, sr, int(rng.choice([1, 2, 2, 3, 4])), int(rng.choice([1, 1, 2])))
            entries.append({"audio_filepath": p, "duration": n / sr, "text": "x"})
        with open(man, "w") as f:
            for e in entries:
                f.write(json.dumps(e) + "\n")
        hyps, _ = asr.transcribe_manifest(man, batch_size=int(rng.choice([1, 7, 32, 64])))
        for e, h in zip(entries, hyps):
            x, sr = audio.read_wav(e["audio_filepath"])
            one = asr.transcribe(x, sample_rate=sr)
            files += 1
            if one != h:
                bad.append((rnd, os.path.basename(e["audio_filepath"]), sr, len(x)))
    print(json.dumps({"rounds": rounds, "files": files, "differences": len(bad), "first": bad[:4], "seconds": round(time.time() - t0, 1)}))


if __name__ == "__main__":
    main()
