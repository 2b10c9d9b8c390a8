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

from conftest import GOLDEN_DIR, ROOT


def _fixture(name):
    path = os.path.join(GOLDEN_DIR, f"thirdparty_{name}.npz")
    if not os.path.exists(path):
        pytest.skip(f"parity unpinned: {os.path.relpath(path)} absent -- run tools/pin_third_party.py where the package is installed")
    return np.load(path, allow_pickle=False)


# ---------------------------------------------------------------- A4: librosa.filters.mel (parts/features.py:199-205)
def test_mel_filterbank_against_librosa():
    from oracle import quartznet_oracle as O
    from viet_asr_amd import frontend_tables
    """BIT equality, not a tolerance: the bank is a constant table.  The shipped default follows librosa's order of roundings
    (float32 triangle store, then the in-place float64 product); if a real librosa equals the single-rounding variant
    instead, the message says so."""
    fb = _fixture("mel")["fb"]
    assert fb.shape == (64, 257) and fb.dtype == np.float32
    args = (16000, 512, 64, 0.0, 8000.0)
    for name, fn in (("oracle", O.slaney_mel_filterbank), ("frontend_tables", frontend_tables.mel_filterbank)):
        match = [v for v in ("librosa", "f64") if np.array_equal(np.asarray(fn(*args, variant=v), dtype=np.float32), fb)]
        worst = {v: int((np.asarray(fn(*args, variant=v), dtype=np.float32) != fb).sum()) for v in ("librosa", "f64")}
        assert "librosa" in match, f"{name}: librosa's bank equals variant(s) {match or 'NONE'}; coefficients that differ: {worst}"
        assert np.array_equal(np.asarray(fn(*args), dtype=np.float32), fb), f"{name}: the default variant is not the pinned one"


# ---------------------------------------------------------------- f3: librosa.load(sr=16000) on 8 kHz input (infer.py:200)
def test_resampler_oracle_against_librosa_load():
    from oracle import audio_oracle as AO
    g = _fixture("resample")
    for sfx in ("", "2"):                 # 8 -> 16 kHz (the reference's case), 11 025 -> 16 000 Hz (ceil(n ratio) != int(n ratio))
        if "pcm" + sfx not in g:
            continue
        x = g["pcm" + sfx].astype(np.float32) / 32768.0
        y = AO.resample(x, int(g["sr_in" + sfx]), int(g["sr_out" + sfx]))
        assert len(y) == len(g["y" + sfx]), (sfx, len(y), len(g["y" + sfx]))      # EXACT length: librosa's fix_length rule
        assert np.abs(y - g["y" + sfx]).max() <= 2e-6


@pytest.mark.gpu
def test_device_resampler_against_librosa_load(gpu):
    from viet_asr_amd import audio
    g = _fixture("resample")
    for sfx in ("", "2"):
        if "pcm" + sfx not in g:
            continue
        pcm = torch.from_numpy(g["pcm" + sfx][None]).to(gpu)
        x = audio.pcm16_to_float(pcm)
        y, n = audio.resample(x, torch.tensor([x.shape[1]], device=gpu), int(g["sr_in" + sfx]), int(g["sr_out" + sfx]))
        assert int(n[0]) == len(g["y" + sfx]), (sfx, int(n[0]), len(g["y" + sfx]))
        assert np.abs(y.cpu().numpy()[0, : int(n[0])] - g["y" + sfx]).max() <= 4e-6


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
    """-> (case, labels, LM path or None, beam width, log-probs).  The LM's file NAME carries the mode the case was pinned in
    (tools/pin_third_party.py BEAM_CASES: "lmK.arpa" = unigram set + character trie, "lmK.txt" = no unigram list): the oracle's
    ``unigrams_for_path`` and the decoder's ``unigrams="auto"`` follow the suffix exactly as build_ctcdecoder does."""
    import sys
    import tempfile
    from viet_asr_amd import configs, synth
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import pin_third_party as pin
    d = tempfile.mkdtemp(prefix="vasr_pin_")
    for k in range(int(g["n_cases"])):
        classes, frames, width, lm_mode, seed = (int(v) for v in g[f"case{k}_meta"])
        if f"case{k}_mode" in g:
            assert str(g[f"case{k}_mode"]) == pin.LM_MODE_NAMES[lm_mode]
        labels = configs.builtin("quartznet15x5" if classes == 29 else "quartznet12x1_vi")["labels"]
        lm_path, ng = pin.lm_file_for(d, k, lm_mode, labels, seed)
        words = sorted(w[0] for w in ng if len(w) == 1 and not w[0].startswith("<"))
        logp = synth.ctc_like_log_probs(1, frames, labels, words, seed=seed)[0]
        yield k, labels, lm_path, width, logp


def test_beam_oracle_against_pyctcdecode():
    from oracle import beam_oracle as BO
    g = _fixture("beam")
    for k, labels, arpa, width, logp in _beam_cases(g):
        lm = BO.LanguageModel(BO.NgramLM.from_arpa(arpa), alpha=0.5, beta=1.5, unigrams=BO.unigrams_for_path(arpa)) if arpa else None
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
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    pin = importlib.import_module("pin_third_party")
    out = str(tmp_path)
    np.savez_compressed(os.path.join(out, "thirdparty_mel.npz"), fb=O.slaney_mel_filterbank(16000, 512, 64, 0.0, 8000.0))
    pcm = np.round(synth.audio_batch(1, 3000, 8)[0][0] * 32767).astype(np.int16)
    pcm2 = np.round(synth.audio_batch(1, 2205, 9)[0][0] * 32767).astype(np.int16)
    np.savez_compressed(os.path.join(out, "thirdparty_resample.npz"), pcm=pcm, sr_in=8000, sr_out=16000,
                        y=AO.resample(pcm.astype(np.float32) / 32768.0, 8000, 16000), pcm2=pcm2, sr_in2=11025, sr_out2=16000,
                        y2=AO.resample(pcm2.astype(np.float32) / 32768.0, 11025, 16000))
    sig, _ = synth.audio_batch(2, 8000, 21, ragged=True)
    np.savez_compressed(os.path.join(out, "thirdparty_stftconv.npz"), batch=2, samples=8000, seed=21, ragged=True,
                        magnitude=O.torch_stft_magnitude(torch.from_numpy(sig), 512, 160, 320).numpy())
    picks = [pin.BEAM_CASES[0], pin.BEAM_CASES[4], pin.BEAM_CASES[6]]            # one of each LM mode: arpa, none, binary
    assert [c[3] for c in picks] == [2, 0, 1]
    store = {"n_cases": len(picks)}
    fake = {"n_cases": len(picks)}
    for k, (classes, frames, width, lm_mode, seed) in enumerate(picks):
        fake[f"case{k}_meta"] = np.array([classes, 40, width, lm_mode, seed])
    for k, labels, arpa, width, logp in _beam_cases(fake):
        lm = BO.LanguageModel(BO.NgramLM.from_arpa(arpa), alpha=0.5, beta=1.5, unigrams=BO.unigrams_for_path(arpa)) if arpa else None
        beams = BO.decode_beams(np.exp(logp.astype(np.float64)).astype(np.float32), labels, width, lm=lm)[:5]
        store[f"case{k}_meta"] = fake[f"case{k}_meta"]
        store[f"case{k}_mode"] = np.array(pin.LM_MODE_NAMES[int(fake[f"case{k}_meta"][3])])
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
