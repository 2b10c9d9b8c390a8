"""Audio ingest + manifest data layer: CPU tests of the host logic, -m gpu tests of the device kernels."""
import json
import os

import numpy as np
import pytest
import torch

from viet_asr_amd import audio
from viet_asr_amd.core import DeviceType, NeuralModuleFactory
from viet_asr_amd.data_layer import AudioToTextDataLayer, word_error_rate

LABELS = list(" abcdefghijklmnopqrstuvwxyz'")


def _tone(n, sr, f=440.0, amp=0.3):
    return (amp * np.sin(2 * np.pi * f * np.arange(n) / sr)).astype(np.float32)


def test_wav_round_trip_and_int_scaling(tmp_path):
    x = _tone(1600, 16000)
    p = str(tmp_path / "a.wav")
    audio.write_wav(p, x, 16000)
    y, sr = audio.read_wav(p)
    assert sr == 16000 and y.dtype == np.float32 and len(y) == 1600
    assert np.abs(y - x).max() <= 1.0 / 32768 + 1e-7                     # int16 quantisation, scaled by 2^-15
    assert np.all(np.round(y * 32768) == np.round(y * 32768).astype(np.int16))


def test_manifest_layer_batches_and_collate(tmp_path):
    NeuralModuleFactory(placement=DeviceType.CPU)
    man = str(tmp_path / "m.json")
    durs = [0.30, 0.10, 0.20, 0.25, 5.0]
    with open(man, "w", encoding="utf-8") as f:
        for i, d in enumerate(durs):
            p = str(tmp_path / f"u{i}.wav")
            audio.write_wav(p, _tone(int(d * 16000), 16000, 200 + 50 * i), 16000)
            f.write(json.dumps({"audio_filepath": p, "duration": d, "text": f"ab c{'de'[i % 2]} Zq"}) + "\n")
    dl = AudioToTextDataLayer(man, LABELS, batch_size=2, max_duration=1.0)
    assert list(dl.output_ports) == ["audio_signal", "a_sig_length", "transcripts", "transcript_length"]
    assert len(dl) == 2 and dl.utterance_order() == [1, 2, 3, 0]          # 5 s clip filtered, bucketed by duration
    batches = list(dl.data_iterator)
    a, al, t, tl = batches[0]
    assert a.shape == (2, 3200) and al.tolist() == [1600, 3200] and not a[0, 1600:].any()   # zero-pad to max
    assert tl.tolist() == [7, 7] and t[0].tolist() == [1, 2, 0, 3, 5, 0, 17]                 # 'Z' dropped, 'q' kept
    assert word_error_rate(["a b c", "x"], ["a b d", "x"]) == pytest.approx(0.25)
    assert word_error_rate(["abc"], ["abd"], use_cer=True) == pytest.approx(1 / 3)
    with pytest.raises(ValueError):
        word_error_rate(["a"], ["a", "b"])


def test_sinc_table_matches_oracle():
    from oracle import audio_oracle as AO
    tab, nt = audio.sinc_table()
    win, nt2 = AO.sinc_window()
    assert nt == nt2 == 512 and tab.shape == (64 * 512 + 1, 2)
    assert np.abs(tab[:, 0] - win).max() < 1e-7 and abs(tab[0, 0] - 0.9475937167399596) < 1e-7


def test_resampler_oracle_length_rule_is_librosas():
    """librosa.load(sr=16000) -> librosa.resample(fix=True): resampy computes int(n * ratio) samples, librosa pads them with
    zeros to ceil(n * ratio), both products in float64 (infer.py:200).  The reference's 8 -> 16 kHz case cannot tell the two
    rules apart; 11 025 -> 16 000 Hz can: 5 000 samples -> 7 256 computed, 7 257 returned."""
    from oracle import audio_oracle as AO
    r = np.random.RandomState(4)
    for sr_in, sr_out, n in ((8000, 16000, 333), (11025, 16000, 5000), (11025, 16000, 300), (16000, 8000, 301), (44100, 16000, 1000)):
        x = r.randn(n).astype(np.float32)
        y = AO.resample(x, sr_in, sr_out)
        ratio = float(sr_out) / sr_in
        assert len(y) == int(np.ceil(n * ratio))
        assert not y[int(n * ratio):].any()                       # the padding fix_length adds
    assert len(AO.resample(np.zeros(5000, dtype=np.float32), 11025, 16000)) == 7257 and int(5000 * (16000.0 / 11025)) == 7256


@pytest.mark.gpu
@pytest.mark.parametrize("sr_in,sr_out", [(8000, 16000), (16000, 8000), (11025, 16000), (16000, 24000), (48000, 16000)])
def test_device_resampler_matches_oracle(gpu, sr_in, sr_out):
    from oracle import audio_oracle as AO
    r = np.random.RandomState(0)
    lens = np.array([3000, 1777], dtype=np.int64)
    x = np.zeros((2, 3000), dtype=np.float32)
    for b in range(2):
        x[b, : lens[b]] = (0.2 * r.randn(lens[b]) + _tone(lens[b], sr_in, 300.0)).astype(np.float32)
    y, ln = audio.resample(torch.from_numpy(x).to(gpu), torch.from_numpy(lens).to(gpu), sr_in, sr_out)
    y, ln = y.cpu().numpy(), ln.cpu().numpy()
    for b in range(2):
        ref = AO.resample(x[b, : lens[b]], sr_in, sr_out)
        assert ln[b] == len(ref)
        assert np.abs(y[b, : ln[b]] - ref).max() < 2e-6
        assert not y[b, ln[b]:].any()
    # a band-limited tone survives 8k -> 16k: compare with the analytic signal away from the edges
    n = 4000
    tone = _tone(n, 8000, 440.0)
    up, _ = audio.resample(torch.from_numpy(tone[None]).to(gpu), torch.tensor([n], device=gpu), 8000, 16000)
    want = _tone(2 * n, 16000, 440.0)
    assert np.abs(up.cpu().numpy()[0, 600:-600] - want[600:-600]).max() < 2e-3


@pytest.mark.gpu
def test_pcm16_to_float_on_device(gpu):
    pcm = torch.from_numpy(np.array([[0, 1, -1, 32767, -32768, 1234, -4321]], dtype=np.int16)).to(gpu)
    out = audio.pcm16_to_float(pcm).cpu().numpy()
    assert out.dtype == np.float32 and np.array_equal(out, pcm.cpu().numpy().astype(np.float32) / 32768.0)
    big = torch.randint(-32768, 32767, (3, 4001), dtype=torch.int16, device=gpu)
    assert torch.equal(audio.pcm16_to_float(big), big.float() / 32768.0)


@pytest.mark.gpu
@pytest.mark.parametrize("batch,samples", [(3, 20000), (40, 32000)])   # the one-frame-per-wavefront STFT form / the 32-frame one
def test_int16_pcm_straight_into_the_front_end(gpu, batch, samples):
    """int16 -> float32 (parts/segment.py:61-74, samples * 2^-15) fused into the front end's staging load
    (vasr_transcribe_greedy_pcm16).  The conversion and the power-of-two product are exact, so every output -- log-probs
    included -- equals the float path's on the converted samples bit for bit; ragged rows, both STFT kernel forms, and the
    host-side batch API (engine.launch) fed with int16 arrays."""
    from viet_asr_amd import configs, synth
    from viet_asr_amd.engine import QuartzNetCTC
    cfg = configs.builtin("quartznet12x1_vi")
    jas = cfg["JasperEncoder"]["jasper"]
    eng = QuartzNetCTC(cfg, synth.encoder_state_dict(jas, 64, 21), synth.decoder_state_dict(1024, len(cfg["labels"]) + 1, 21), device=gpu)
    sig, lens = synth.audio_batch(batch, samples, 21, ragged=True)
    pcm = np.clip(np.round(sig * 32768.0 * 4), -32768, 32767).astype(np.int16)
    for b in range(batch):
        pcm[b, lens[b]:] = 0
    d_pcm, d_len = torch.from_numpy(pcm).to(gpu), torch.from_numpy(lens).to(gpu)
    a = eng.forward(d_pcm, d_len, want_logp=True)
    b_ = eng.forward(audio.pcm16_to_float(d_pcm), d_len, want_logp=True)
    def same_ids(p, q):      # (ids past id_len are unwritten)
        n = p["id_len"].cpu().numpy()
        return torch.equal(p["id_len"], q["id_len"]) and all(torch.equal(p["ids"][r, : n[r]], q["ids"][r, : n[r]]) for r in range(batch))
    for k in ("pred", "enc_len", "logp"):
        assert torch.equal(a[k], b_[k]), k
    assert same_ids(a, b_) and int(a["id_len"].max()) > 0
    for ri in (False, True):
        a = eng.forward(d_pcm, d_len, row_independent=ri)
        b_ = eng.forward(audio.pcm16_to_float(d_pcm), d_len, row_independent=ri)
        assert same_ids(a, b_)
    eng.forward(d_pcm, d_len)      # (leave the handle in the batched mode)
    got = eng.launch([pcm[i, : lens[i]] for i in range(batch)]).texts()
    want = eng.launch([(pcm[i, : lens[i]].astype(np.float32) / 32768.0) for i in range(batch)]).texts()
    assert got == want and any(got)
    with pytest.raises(ValueError):
        eng.forward(d_pcm.to(torch.int32), d_len)


@pytest.mark.gpu
def test_randomised_resampler_and_greedy_tail_cases(gpu):
    """Forty cases each of tests/devtools/fuzz_audio.py: random rate pairs / batch shapes / ragged lengths through the
    resampler, random shapes and class counts (with exact ties) through argmax + CTC collapse."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "devtools"))
    import fuzz_audio
    bad = [m for fn in (fuzz_audio.resample_case, fuzz_audio.greedy_case) for m in (fn(c) for c in range(40)) if m]
    assert not bad, bad


@pytest.mark.gpu
def test_manifest_transcription_equals_file_by_file_calls(gpu, tmp_path):
    """VietASR.transcribe_manifest: duration-sorted, batched, pipelined and row-independent -- the transcripts of the
    reference-style one-file-at-a-time loop (infer.py:194-206), in manifest order, and a WER of 0 against them."""
    from viet_asr_amd import configs, synth
    from viet_asr_amd.infer import VietASR
    cfg = configs.builtin("quartznet12x1_vi")
    jas = cfg["JasperEncoder"]["jasper"]
    enc_p, dec_p = str(tmp_path / "JasperEncoder-STEP-1.pt"), str(tmp_path / "JasperDecoderForCTC-STEP-1.pt")
    torch.save({k: torch.as_tensor(v) for k, v in synth.encoder_state_dict(jas, 64, 3).items()}, enc_p)
    torch.save({k: torch.as_tensor(v) for k, v in synth.decoder_state_dict(1024, 91, 3).items()}, dec_p)
    asr = VietASR("quartznet12x1_vi", enc_p, dec_p, device="gpu", decoder="greedy")
    rng = np.random.default_rng(8)
    lens = [9000, 31000, 16000, 16000, 4000, 25000, 8000, 12000, 20000]
    rates = [16000] * 8 + [8000]
    paths = []
    for i, (n, sr) in enumerate(zip(lens, rates)):
        p = str(tmp_path / f"u{i}.wav")
        audio.write_wav(p, (0.1 * rng.standard_normal(n)).astype(np.float32), sr)
        paths.append(p)
    alone = [asr.transcribe(*audio.read_wav(p)) for p in paths]            # transcribe(signal, sample_rate)
    man = str(tmp_path / "m.json")
    with open(man, "w", encoding="utf-8") as f:
        for p, n, sr, t in zip(paths, lens, rates, alone):
            f.write(json.dumps({"audio_filepath": p, "duration": n / sr, "text": t}, ensure_ascii=False) + "\n")
    hyps, wer = asr.transcribe_manifest(man, batch_size=4)
    assert hyps == alone and wer == 0.0


@pytest.mark.gpu
def test_randomised_front_end_cases(gpu):
    """Forty cases of tests/devtools/fuzz_frontend.py: batch shapes, ragged lengths, exact hop multiples, rows of 1-3
    frames (NaN statistics like the reference), silent and loud rows.  Log-mel before normalisation within MEL_TOL,
    normalised features within what (x - mean) / (std + 1e-5) makes of that error, NaN pattern and masks identical."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "devtools"))
    import fuzz_frontend
    bad = [m for m in (fuzz_frontend.frontend_case(c) for c in range(40)) if m]
    assert not bad, bad
