"""Third-party arithmetic on the hot path: pinned when `tools/pin_third_party.py` has been run on a machine that has the
packages (fixtures tests/golden/thirdparty_*.npz), otherwise skipped with the words "parity unpinned".

SURVEY section 8 rows A4 (librosa mel filterbank), A12 (pyctcdecode + kenlm beam search), f3 (resampy kaiser_best behind
librosa.load) and the `stft_conv=True` STFT (torch_stft) live in pip packages that are neither in /root/reference nor in the
build image.  Until the fixtures exist the oracle's restatements of their PUBLISHED algorithms are the only pin; once they
do, every test below compares the CPU oracle (no GPU needed) and the device kernels (-m gpu) against the packages' own output.
"""
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN_DIR


def _fixture(name):
    path = os.path.join(GOLDEN_DIR, f"thirdparty_{name}.npz")
    if not os.path.exists(path):
        pytest.skip(f"parity unpinned: {os.path.relpath(path)} absent -- run tools/pin_third_party.py where the package is installed")
    return np.load(path, allow_pickle=False)


# ---------------------------------------------------------------- A4: librosa.filters.mel (parts/features.py:199-205)
def test_mel_filterbank_against_librosa():
    from oracle import quartznet_oracle as O
    from viet_asr_amd import frontend_tables
    fb = _fixture("mel")["fb"]
    assert fb.shape == (64, 257)
    for ours in (O.slaney_mel_filterbank(16000, 512, 64, 0.0, 8000.0), frontend_tables.mel_filterbank(16000, 512, 64, 0.0, 8000.0)):
        assert np.abs(np.asarray(ours, dtype=np.float32) - fb).max() <= 1e-8 + 1e-6 * np.abs(fb).max()


# ---------------------------------------------------------------- f3: librosa.load(sr=16000) on 8 kHz input (infer.py:200)
def test_resampler_oracle_against_librosa_load():
    from oracle import audio_oracle as AO
    g = _fixture("resample")
    x = g["pcm"].astype(np.float32) / 32768.0
    y = AO.resample(x, int(g["sr_in"]), int(g["sr_out"]))
    assert len(y) == len(g["y"]) and np.abs(y - g["y"]).max() <= 2e-6


@pytest.mark.gpu
def test_device_resampler_against_librosa_load(gpu):
    from viet_asr_amd import audio
    g = _fixture("resample")
    pcm = torch.from_numpy(g["pcm"][None]).to(gpu)
    x = audio.pcm16_to_float(pcm)
    y, n = audio.resample(x, torch.tensor([x.shape[1]], device=gpu), int(g["sr_in"]), int(g["sr_out"]))
    assert int(n[0]) == len(g["y"]) and np.abs(y.cpu().numpy()[0, : int(n[0])] - g["y"]).max() <= 4e-6


# ---------------------------------------------------------------- stft_conv=True: torch_stft.STFT (parts/features.py:155-166)
def _stft_inputs(g):
    from viet_asr_amd import synth
    return synth.audio_batch(int(g["batch"]), int(g["samples"]), int(g["seed"]), bool(g["ragged"]))


def test_stft_conv_oracle_against_torch_stft():
    from oracle import quartznet_oracle as O
    g = _fixture("stftconv")
    sig, _ = _stft_inputs(g)
    mag = O.torch_stft_magnitude(torch.from_numpy(sig), 512, 160, 320).numpy()
    assert mag.shape == g["magnitude"].shape
    assert np.abs(mag - g["magnitude"]).max() <= 2e-5 * max(1.0, np.abs(g["magnitude"]).max())


# ---------------------------------------------------------------- A12: pyctcdecode + kenlm (beam_search_decoder.py:82-102)
def _beam_cases(g):
    from viet_asr_amd import configs, synth
    import tempfile
    d = tempfile.mkdtemp(prefix="vasr_pin_")
    for k in range(int(g["n_cases"])):
        classes, frames, width, with_lm, seed = (int(v) for v in g[f"case{k}_meta"])
        labels = configs.builtin("quartznet15x5" if classes == 29 else "quartznet12x1_vi")["labels"]
        arpa = os.path.join(d, f"lm{k}.arpa")
        ng = synth.synthetic_arpa(arpa, labels, n_words=2000, n_bigrams=4000, n_trigrams=4000, seed=seed)
        words = sorted(w[0] for w in ng if len(w) == 1 and not w[0].startswith("<"))
        logp = synth.ctc_like_log_probs(1, frames, labels, words, seed=seed)[0]
        yield k, labels, (arpa if with_lm else None), width, logp


def test_beam_oracle_against_pyctcdecode():
    from oracle import beam_oracle as BO
    g = _fixture("beam")
    for k, labels, arpa, width, logp in _beam_cases(g):
        lm = BO.LanguageModel(BO.NgramLM.from_arpa(arpa), alpha=0.5, beta=1.5) if arpa else None
        probs = np.exp(logp.astype(np.float64)).astype(np.float32)
        ref = BO.decode_beams(probs, labels, width, lm=lm)[:5]
        want_text, want_lm = [str(t) for t in g[f"case{k}_text"]], g[f"case{k}_lm_score"]
        assert ref[0][0] == str(g[f"case{k}_decode"]) == want_text[0], (k, ref[0][0], want_text[0])
        assert [r[0] for r in ref] == want_text[: len(ref)], k
        assert np.abs(np.array([r[2] for r in ref]) - want_lm[: len(ref)]).max() <= 1e-3, k


@pytest.mark.gpu
def test_device_beam_search_against_pyctcdecode(gpu):
    from viet_asr_amd.beam import BeamSearchDecoder
    g = _fixture("beam")
    for k, labels, arpa, width, logp in _beam_cases(g):
        dec = BeamSearchDecoder(labels, lm_path=arpa, alpha=0.5, beta=1.5)
        ids, n, score = dec.decode_ids(torch.from_numpy(logp[None]).to(gpu), width)
        text = "".join(labels[c] for c in ids[0, : int(n[0])].tolist())
        assert text == str(g[f"case{k}_decode"]), (k, text)
        assert abs(float(score[0]) - float(g[f"case{k}_lm_score"][0])) <= 2e-3 * max(1.0, abs(float(score[0])) / 50), k


# ---------------------------------------------------------------- the harness itself
def test_pin_harness_with_oracle_stand_ins(tmp_path, monkeypatch):
    """The fixtures cannot exist in this image; so that the FIRST real ones do not trip over the harness, write stand-ins
    in the same format from the oracle's own outputs (what tools/pin_third_party.py writes, the package replaced by its
    restatement) and run the CPU comparisons above against them.  Says nothing about parity -- it tests the plumbing."""
    import sys
    import importlib
    from oracle import audio_oracle as AO, beam_oracle as BO, quartznet_oracle as O
    from viet_asr_amd import synth
    sys.path.insert(0, os.path.join(os.path.dirname(GOLDEN_DIR), "..", "tools"))
    pin = importlib.import_module("pin_third_party")
    out = str(tmp_path)
    np.savez_compressed(os.path.join(out, "thirdparty_mel.npz"), fb=O.slaney_mel_filterbank(16000, 512, 64, 0.0, 8000.0))
    pcm = np.round(synth.audio_batch(1, 3000, 8)[0][0] * 32767).astype(np.int16)
    np.savez_compressed(os.path.join(out, "thirdparty_resample.npz"), pcm=pcm, sr_in=8000, sr_out=16000,
                        y=AO.resample(pcm.astype(np.float32) / 32768.0, 8000, 16000))
    sig, _ = synth.audio_batch(2, 8000, 21, ragged=True)
    np.savez_compressed(os.path.join(out, "thirdparty_stftconv.npz"), batch=2, samples=8000, seed=21, ragged=True,
                        magnitude=O.torch_stft_magnitude(torch.from_numpy(sig), 512, 160, 320).numpy())
    store = {"n_cases": 2}
    fake = {"n_cases": 2}
    for k, (classes, frames, width, with_lm, seed) in enumerate(pin.BEAM_CASES[:2]):
        fake[f"case{k}_meta"] = np.array([classes, 40, width, int(with_lm), seed])
    for k, labels, arpa, width, logp in _beam_cases(fake):
        lm = BO.LanguageModel(BO.NgramLM.from_arpa(arpa), alpha=0.5, beta=1.5) if arpa else None
        beams = BO.decode_beams(np.exp(logp.astype(np.float64)).astype(np.float32), labels, width, lm=lm)[:5]
        store[f"case{k}_meta"] = fake[f"case{k}_meta"]
        store[f"case{k}_text"] = np.array([b[0] for b in beams])
        store[f"case{k}_logit_score"] = np.array([b[1] for b in beams])
        store[f"case{k}_lm_score"] = np.array([b[2] for b in beams])
        store[f"case{k}_decode"] = np.array(beams[0][0])
    np.savez_compressed(os.path.join(out, "thirdparty_beam.npz"), **store)
    monkeypatch.setattr(sys.modules[__name__], "GOLDEN_DIR", out)
    test_mel_filterbank_against_librosa()
    test_resampler_oracle_against_librosa_load()
    test_stft_conv_oracle_against_torch_stft()
    test_beam_oracle_against_pyctcdecode()
