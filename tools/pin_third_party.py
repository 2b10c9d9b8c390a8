#!/usr/bin/env python3
"""Pin the four pieces of THIRD-PARTY arithmetic on the viet-asr hot path that this repository can only restate.

The reference computes them inside pip packages that are neither vendored in /root/reference nor installed in the build
image (no network): SURVEY section 8 rows A4, A12, f3 and the `stft_conv=True` front end are therefore "parity unpinned".
Run this script ONCE on a machine that has the packages -- e.g. `pip install "librosa<0.10" resampy torch_stft pyctcdecode
kenlm` -- and commit what it writes: from then on `tests/test_thirdparty_pins.py` compares BOTH the CPU oracle and the
device kernels with the packages' own outputs instead of skipping with "parity unpinned".

    python tools/pin_third_party.py [--out tests/golden] [--only mel,resample,stftconv,beam]

fixture                                 produced by (reference call site)                              pins
tests/golden/thirdparty_mel.npz         librosa.filters.mel(16000, 512, n_mels=64, fmin=0, fmax=8000)  oracle.slaney_mel_filterbank,
                                        (nemo/collections/asr/parts/features.py:199-205)               frontend_tables (row A4)
tests/golden/thirdparty_resample.npz    librosa.load(path, sr=16000) on an 8 kHz file = resampy        oracle/audio_oracle.resample,
                                        kaiser_best (infer.py:200, app.py:66,82)                       vasr_resample_f32 (row f3)
tests/golden/thirdparty_stftconv.npz    torch_stft.STFT(512, 160, 320, "hann").transform(x)[0]         oracle.torch_stft_magnitude,
                                        (parts/features.py:155-166)                                    the stft_conv=True front end
tests/golden/thirdparty_beam.npz        pyctcdecode.build_ctcdecoder(vocab, kenlm_model_path=arpa,     oracle/beam_oracle.decode_beams,
                                        alpha, beta).decode_beams(probs, beam_width)                   vasr_beam_search_f32 (row A12)
                                        (nemo/collections/asr/beam_search_decoder.py:82-102)

Inputs are regenerated from seeds by the tests (viet_asr_amd.synth), except where the package output depends on an input
file: those inputs are stored in the fixture.  Only numpy arrays are written -- data, no package source.
"""
import argparse
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

# (classes incl. blank, frames, beam width, LM mode, seed) -- the reference's own widths: 20 ctor, 50 app.py, 100 CLI.
# LM mode: 0 = no LM; 2 = the model handed over as "lmK.arpa" -- build_ctcdecoder then reads the file's unigrams and builds its
# character trie (oracle/beam_oracle.py header); 1 = the SAME ARPA text as "lmK.txt": kenlm reads it all the same (it sniffs
# the content), but pyctcdecode loads unigrams for the ".arpa" suffix only, so this is the behaviour the reference got from
# its `3-gram-lm.binary` (no unigram list).  Both are pinned; `caseK_mode` records which one a case used.
LM_MODE_NAMES = ["none", "binary", "arpa"]
BEAM_CASES = [
    (29, 120, 20, 2, 1), (29, 200, 50, 2, 2), (29, 200, 100, 2, 3), (29, 150, 128, 2, 4),
    (29, 150, 50, 0, 5), (91, 120, 100, 2, 6),
    (29, 120, 20, 1, 1), (29, 200, 100, 1, 3), (91, 120, 100, 1, 6),
]


def lm_file_for(d, k, lm_mode, labels, seed):
    """Writes case k's synthetic 3-gram model under the name its mode asks for; -> (path or None, n-grams)."""
    from viet_asr_amd import synth
    path = os.path.join(d, f"lm{k}" + (".arpa" if lm_mode != 1 else ".txt"))
    ng = synth.synthetic_arpa(path, labels, n_words=2000, n_bigrams=4000, n_trigrams=4000, seed=seed)
    return (path if lm_mode else None), ng


def versions(*mods):
    out = {}
    for m in mods:
        try:
            out[m] = getattr(__import__(m), "__version__", "?")
        except Exception:  # noqa: BLE001
            out[m] = "absent"
    return out


def pin_mel(out):
    import librosa
    try:
        fb = librosa.filters.mel(16000, 512, n_mels=64, fmin=0, fmax=8000)            # the reference's positional call (< 0.10)
    except TypeError:
        fb = librosa.filters.mel(sr=16000, n_fft=512, n_mels=64, fmin=0, fmax=8000)   # same function, keyword-only API
    np.savez_compressed(os.path.join(out, "thirdparty_mel.npz"), fb=np.asarray(fb, dtype=np.float32),
                        versions=str(versions("librosa", "numpy")))
    from oracle.quartznet_oracle import slaney_mel_filterbank
    same = [v for v in ("librosa", "f64") if np.array_equal(slaney_mel_filterbank(16000, 512, 64, 0.0, 8000.0, variant=v), fb)]
    print("mel", fb.shape, fb.dtype, float(fb.sum()), float(fb.max()), "| equals the restatement's variant(s):", same or "NONE")


def pin_resample(out):
    import librosa
    import soundfile as sf
    r = np.random.RandomState(8)
    x = r.uniform(-0.3, 0.3, 12000).astype(np.float32)
    x = (0.25 * np.roll(x, 1) + 0.5 * x + 0.25 * np.roll(x, -1)).astype(np.float32)
    pcm = np.round(x * 32767).astype(np.int16)
    d = tempfile.mkdtemp(prefix="vasr_pin_")
    path = os.path.join(d, "pin8k.wav")
    sf.write(path, pcm, 8000, subtype="PCM_16")
    y, sr = librosa.load(path, sr=16000)                                                # infer.py:200
    # a second file at 11 025 Hz: ceil(n * ratio) != int(n * ratio) there (5 000 samples -> 7 256 computed, 7 257 returned),
    # which pins librosa's fix_length rule around resampy's int(n * ratio) output -- at 8 -> 16 kHz the two rules cannot be told apart
    pcm2 = pcm[:5000]
    path2 = os.path.join(d, "pin11k.wav")
    sf.write(path2, pcm2, 11025, subtype="PCM_16")
    y2, sr2 = librosa.load(path2, sr=16000)
    np.savez_compressed(os.path.join(out, "thirdparty_resample.npz"), pcm=pcm, y=np.asarray(y, dtype=np.float32), sr_in=8000,
                        sr_out=int(sr), pcm2=pcm2, y2=np.asarray(y2, dtype=np.float32), sr_in2=11025, sr_out2=int(sr2),
                        versions=str(versions("librosa", "resampy", "soundfile")))
    print("resample", pcm.shape, "->", y.shape, "|", pcm2.shape, "->", y2.shape)


def pin_stftconv(out):
    import torch
    from torch_stft import STFT
    from viet_asr_amd import synth
    sig, lens = synth.audio_batch(2, 8000, 21, ragged=True)
    stft = STFT(512, 160, 320, "hann")                                                  # parts/features.py:166
    with torch.no_grad():
        mag = stft.transform(torch.from_numpy(sig))[0]
    np.savez_compressed(os.path.join(out, "thirdparty_stftconv.npz"), batch=2, samples=8000, seed=21, ragged=True,
                        magnitude=mag.numpy().astype(np.float32), versions=str(versions("torch_stft", "torch", "librosa")))
    print("stftconv", tuple(mag.shape))


def pin_beam(out):
    from pyctcdecode import build_ctcdecoder
    from viet_asr_amd import configs, synth
    store = {"versions": str(versions("pyctcdecode", "kenlm")), "n_cases": len(BEAM_CASES)}
    d = tempfile.mkdtemp(prefix="vasr_pin_")
    for k, (classes, frames, width, lm_mode, seed) in enumerate(BEAM_CASES):
        labels = configs.builtin("quartznet15x5" if classes == 29 else "quartznet12x1_vi")["labels"]
        lm_path, ng = lm_file_for(d, k, lm_mode, labels, seed)
        words = sorted(w[0] for w in ng if len(w) == 1 and not w[0].startswith("<"))
        logp = synth.ctc_like_log_probs(1, frames, labels, words, seed=seed)[0]
        probs = np.exp(logp.astype(np.float64)).astype(np.float32)                      # what the reference hands over (:97)
        dec = build_ctcdecoder(list(labels), kenlm_model_path=lm_path, alpha=0.5, beta=1.5)   # the reference's call (:82-87)
        beams = dec.decode_beams(probs, beam_width=width)[:5]
        store[f"case{k}_meta"] = np.array([classes, frames, width, lm_mode, seed])
        store[f"case{k}_mode"] = np.array(LM_MODE_NAMES[lm_mode])
        store[f"case{k}_text"] = np.array([b[0] for b in beams])
        store[f"case{k}_logit_score"] = np.array([b[-2] for b in beams], dtype=np.float64)
        store[f"case{k}_lm_score"] = np.array([b[-1] for b in beams], dtype=np.float64)
        store[f"case{k}_decode"] = np.array(dec.decode(probs, beam_width=width))        # the call the reference makes (:98-101)
        print("beam", k, repr(beams[0][0][:50]), beams[0][-1])
    np.savez_compressed(os.path.join(out, "thirdparty_beam.npz"), **store)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden"))
    ap.add_argument("--only", default="mel,resample,stftconv,beam")
    a = ap.parse_args()
    todo = {"mel": pin_mel, "resample": pin_resample, "stftconv": pin_stftconv, "beam": pin_beam}
    failed = []
    for name in a.only.split(","):
        try:
            todo[name](a.out)
        except ImportError as e:
            failed.append(name)
            print(f"{name}: NOT pinned -- {e} (install the package and run again)")
    sys.exit(1 if failed else 0)


if __name__ == "__main__":
    main()
