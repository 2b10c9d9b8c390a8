#!/usr/bin/env python3
"""The reference-style path against the fused path (dev tool, GPU): python tests/devtools/fuzz_dag.py [n] [seed0] [max_seconds]

VietASR.transcribe(x) runs infer.py's DAG -- data layer -> AudioToMelSpectrogramPreprocessor -> JasperEncoder -> JasperDecoderForCTC
-> GreedyCTCDecoder as separate NeuralModules through NeuralModuleFactory.infer, port tensors between them -- one utterance per
call like the reference.  transcribe_batch(..., row_independent=True) runs the fused one-call path on a batch and promises, per
row, "what transcribe returns for that signal alone".  The two paths share no entry point below Python (vasr_melspec_f32 /
vasr_encoder_f32 / vasr_decoder_logsoftmax_f32 / vasr_greedy_argmax against vasr_transcribe_greedy_*), so the promise is checked
here on random signals: lengths 0.2-12 s, three levels, float and int16 PCM, 16 and 8 kHz, batches of 1-24 for the fused side."""
import os
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import viet_asr_amd  # noqa: E402,F401
from viet_asr_amd import configs, synth  # noqa: E402
from viet_asr_amd.infer import VietASR  # noqa: E402

_ASR = {}
STATS = {"cases": 0, "signals": 0, "differ": 0, "chars": 0}


def asr_for(model):
    if model not in _ASR:
        cfg = configs.builtin(model)
        jas = cfg["JasperEncoder"]["jasper"]
        d = tempfile.mkdtemp(prefix="vasr_dag_")
        enc_p, dec_p = os.path.join(d, "JasperEncoder-STEP-1.pt"), os.path.join(d, "JasperDecoderForCTC-STEP-1.pt")
        torch.save({k: torch.as_tensor(v) for k, v in synth.encoder_state_dict(jas, 64, 8).items()}, enc_p)
        torch.save({k: torch.as_tensor(v) for k, v in synth.decoder_state_dict(jas[-1]["filters"], len(cfg["labels"]) + 1, 8).items()}, dec_p)
        _ASR[model] = VietASR(model, enc_p, dec_p, device="gpu", decoder="greedy")
    return _ASR[model]


_BEAM = {}


def beam_asr_for(model):
    if model not in _BEAM:
        cfg = configs.builtin(model)
        jas = cfg["JasperEncoder"]["jasper"]
        d = tempfile.mkdtemp(prefix="vasr_dagb_")
        enc_p, dec_p = os.path.join(d, "JasperEncoder-STEP-1.pt"), os.path.join(d, "JasperDecoderForCTC-STEP-1.pt")
        torch.save({k: torch.as_tensor(v) for k, v in synth.encoder_state_dict(jas, 64, 8).items()}, enc_p)
        torch.save({k: torch.as_tensor(v) for k, v in synth.decoder_state_dict(jas[-1]["filters"], len(cfg["labels"]) + 1, 8).items()}, dec_p)
        arpa = os.path.join(d, "lm.arpa")
        synth.synthetic_arpa(arpa, cfg["labels"], n_words=3000, n_bigrams=8000, n_trigrams=8000, seed=8)
        _BEAM[model] = VietASR(model, enc_p, dec_p, device="gpu", decoder="beam", lm_path=arpa, beam_width=16)
    return _BEAM[model]


def dag_beam_case(case):
    """The beam wiring (infer.py:132-160: BeamSearchDecoderWithLM behind the decoder module, one utterance per call) against
    transcribe_batch(decoder="beam", row_independent=True), which promises "the transcripts of batch-1 calls"."""
    rng = np.random.default_rng(400000 + case)
    asr = beam_asr_for("quartznet12x1_vi")
    B = int(rng.choice([1, int(rng.integers(2, 7)), int(rng.integers(7, 20))]))
    sigs = [(float(rng.choice([0.004, 0.06, 0.5])) * rng.standard_normal(int(rng.integers(3200, 64000)))).astype(np.float32) for _ in range(B)]
    fused = asr.transcribe_batch(sigs, decoder="beam", row_independent=True)
    STATS["cases"] += 1
    for i, s in enumerate(sigs):
        one = asr.transcribe(s)
        STATS["signals"] += 1
        STATS["chars"] += len(one)
        if one != fused[i]:
            STATS["differ"] += 1
            return f"dag beam case {case}: row {i} of {B} (n {len(s)}): DAG {one[:40]!r} vs batch {fused[i][:40]!r}"
    return None


def dag_case(case):
    rng = np.random.default_rng(300000 + case)
    asr = asr_for(str(rng.choice(["quartznet12x1_vi", "quartznet12x1_vi", "quartznet15x5"])))
    B = int(rng.choice([1, int(rng.integers(2, 7)), int(rng.integers(7, 25))]))
    rate = int(rng.choice([16000, 16000, 8000]))
    pcm = bool(rng.random() < 0.35)
    sigs = []
    for _ in range(B):
        n = int(rng.integers(int(0.2 * rate), int((12 if B < 7 else 4) * rate)))
        x = float(rng.choice([0.004, 0.06, 0.5])) * rng.standard_normal(n)
        sigs.append(np.clip(x * 32768, -32768, 32767).astype(np.int16) if pcm else x.astype(np.float32))
    fused = asr.transcribe_batch(sigs, sample_rate=rate, row_independent=True)
    STATS["cases"] += 1
    for i, s in enumerate(sigs):
        one = asr.transcribe(s, sample_rate=rate)
        STATS["signals"] += 1
        STATS["chars"] += len(one)
        if one != fused[i]:
            STATS["differ"] += 1
            k = next((j for j in range(min(len(one), len(fused[i]))) if one[j] != fused[i][j]), min(len(one), len(fused[i])))
            return (f"dag case {case}: row {i} of {B} (n {len(s)}, rate {rate}, pcm {pcm}): DAG {one[max(0, k - 8):k + 8]!r} vs fused "
                    f"{fused[i][max(0, k - 8):k + 8]!r} at character {k} of {len(one)} / {len(fused[i])}")
    return None


if __name__ == "__main__":
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    S0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    LIMIT = float(sys.argv[3]) if len(sys.argv) > 3 else 1e9
    fn = dag_beam_case if len(sys.argv) > 4 and sys.argv[4] == "beam" else dag_case
    t0, bad = time.time(), 0
    for case in range(S0, S0 + N):
        if time.time() - t0 > LIMIT:
            break
        msg = fn(case)
        if msg:
            bad += 1
            if bad <= 12:
                print("MISMATCH", msg, flush=True)
    print(f"{STATS['cases']} cases from {S0} ({STATS['signals']} signals, {STATS['chars']} characters), {bad} mismatching cases, {time.time() - t0:.0f} s")
