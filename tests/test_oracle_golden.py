"""The oracle is pinned against outputs of the real reference modules (tests/golden/make_golden.py)."""
import numpy as np
import pytest

from conftest import BAND_CASES, DITHER_CASES, GOLDEN_CASES, OPTION_CASES, REAL_AUDIO_CASES, load_golden, load_real_audio_golden, oracle_pre_kwargs
from oracle import audio_oracle as AO
from oracle import quartznet_oracle as O


@pytest.mark.parametrize("name", GOLDEN_CASES + BAND_CASES + OPTION_CASES + DITHER_CASES)
def test_oracle_matches_reference_outputs(name):
    g, cfg, sig, lens, enc_sd, dec_sd = load_golden(name)
    assert (lens == g["lens"]).all()
    if name in DITHER_CASES:
        import torch
        torch.manual_seed(int(g["seed"]))      # make_golden.py: the reference drew its noise from the CPU generator under this seed
    r = O.forward_all(sig, lens, enc_sd, dec_sd, cfg["JasperEncoder"]["jasper"], **oracle_pre_kwargs(cfg))
    # same ATen primitives as the reference -> bit-exact on the same host; 2e-6 leaves room for
    # a different CPU dispatch (AVX2 vs AVX512 kernels) on another box
    assert np.abs(O.slaney_mel_filterbank() - g["fb"]).max() == 0.0
    assert np.abs(r["mel"].numpy() - g["mel"]).max() <= 2e-5
    assert (r["seq"].numpy() == g["seq"]).all()
    assert r["enc_len"].dtype.is_floating_point and (r["enc_len"].numpy() == g["enc_len"]).all()   # quirk Q3
    assert np.abs(r["enc"][:, ::37, ::5].numpy() - g["enc_slice"]).max() <= 1e-4
    assert abs(float(r["enc"].double().sum()) - float(g["enc_sum"])) <= 1e-6 * float(g["enc_abs_sum"])
    assert np.abs(r["logp"].numpy() - g["logp"]).max() <= 2e-4
    assert (r["pred"].numpy() == g["pred"]).all()
    assert O.ctc_decode_strings(r["pred"], cfg["labels"]) == [str(s) for s in g["hyp"]]
    if "logp_syn" in g.files:   # band-limited vi case: the same encoder output through the seeded head
        from viet_asr_amd import synth
        syn = synth.decoder_state_dict(1024, len(cfg["labels"]) + 1, int(g["seed"]))
        r2 = O.forward_all(sig, lens, enc_sd, syn, cfg["JasperEncoder"]["jasper"])
        assert np.abs(r2["logp"].numpy() - g["logp_syn"]).max() <= 2e-4
        assert (r2["pred"].numpy() == g["pred_syn"]).all()
        assert O.ctc_decode_strings(r2["pred"], cfg["labels"]) == [str(s) for s in g["hyp_syn"]] and len(str(g["hyp_syn"][0])) > 10


def test_band_limited_fixtures_are_band_limited():
    """The round-5 fixtures really are in the regime they are named after: nothing above 4 kHz but the filter's floor,
    i.e. the upper mel bins (no weight below 4.2 kHz: the last 12 of 64) see only that floor."""
    for name in BAND_CASES:
        g, cfg, sig, lens, enc_sd, dec_sd = load_golden(name)
        spec = np.abs(np.fft.rfft(sig[0, : lens[0]].astype(np.float64))) ** 2
        f = np.fft.rfftfreq(int(lens[0]), 1 / 16000.0)
        assert spec[f > 4400].sum() < 1e-5 * spec[f < 3600].sum()      # (what is left is the leakage of the row's own edges)
        fb = g["fb"]
        upper = [m for m in range(64) if fb[m, : int(4200 / 8000 * 256)].sum() == 0]
        assert len(upper) >= 12


@pytest.mark.parametrize("name", REAL_AUDIO_CASES)
def test_oracle_matches_reference_on_real_recordings(name):
    """BASELINE config 1 plumbing: a 16 kHz and an 8 kHz recording of the reference's audio_samples/, decoded the way
    infer.py:200 does (PCM / 2^15; 8 kHz -> 16 kHz by the resampy restatement), one utterance per call."""
    g, cfg, pcm, sr, enc_sd, real_head, syn_head = load_real_audio_golden(name)
    x = pcm.astype(np.float32) / 32768.0
    if sr != 16000:
        x = AO.resample(x, sr, 16000)
    assert len(x) == int(g["samples16"])
    jas = cfg["JasperEncoder"]["jasper"]
    r = O.forward_all(x[None], np.array([len(x)]), enc_sd, real_head, jas)
    assert (r["seq"].numpy() == g["seq"]).all() and (r["enc_len"].numpy() == g["enc_len"]).all()
    assert abs(float(r["mel"].double().sum()) - float(g["mel_sum"])) <= 1e-3
    assert np.abs(r["mel"][:, ::7, ::11].numpy() - g["mel_slice"]).max() <= 2e-5
    assert np.abs(r["logp"].numpy() - g["logp"]).max() <= 2e-4
    assert (r["pred"].numpy() == g["pred"]).all()
    assert O.ctc_decode_strings(r["pred"], cfg["labels"]) == [str(s) for s in g["hyp"]]
    r2 = O.forward_all(x[None], np.array([len(x)]), enc_sd, syn_head, jas)
    assert np.abs(r2["logp"].numpy() - g["logp_syn"]).max() <= 2e-4
    assert O.ctc_decode_strings(r2["pred"], cfg["labels"]) == [str(s) for s in g["hyp_syn"]] and len(str(g["hyp_syn"][0])) > 20


def test_quirks_are_in_the_goldens():
    g = np.load(__import__("os").path.join(__import__("conftest").GOLDEN_DIR, "vi12x1_b3_ragged.npz"))
    # Q3: stride-2 length arithmetic, float encoded lengths
    assert g["enc_len"].dtype == np.float32
    assert list(g["seq"]) == [int(np.ceil(l / 160)) for l in g["lens"]]
    assert list(g["enc_len"]) == [float((s - 1) // 2 + 1) for s in g["seq"]]
    # Q2: len % 160 == 0 -> T = seq + 1
    g2 = np.load(__import__("os").path.join(__import__("conftest").GOLDEN_DIR, "vi12x1_b2_q2_realdec.npz"))
    assert g2["mel"].shape[2] == int(g2["seq"].max()) + 1
    # Q4: padded frames of the shorter utterance are decoded (non-blank predictions past enc_len)
    assert g["pred"].shape[1] == int(max(g["enc_len"]))


def test_torch_stft_restatement_is_the_periodic_window_dft():
    """stft_conv=True (PARITY UNPINNED, torch_stft absent): the oracle restates the package as a conv1d with a windowed
    Fourier basis; an independent formulation -- torch.stft with the periodic window -- must give the same magnitudes,
    and the symmetric window (stft_conv=False) must not."""
    import torch
    from oracle import quartznet_oracle as O
    x = torch.randn(3, 5000, generator=torch.Generator().manual_seed(0))
    m = O.torch_stft_magnitude(x, 512, 160, 320)
    per = torch.stft(x, 512, 160, 320, window=torch.hann_window(320, periodic=True), center=True, return_complex=True,
                     pad_mode="reflect").abs()
    sym = torch.stft(x, 512, 160, 320, window=torch.hann_window(320, periodic=False), center=True, return_complex=True,
                     pad_mode="reflect").abs()
    assert m.shape == per.shape == (3, 257, 32)
    assert float((m - per).abs().max()) < 1e-4 * float(per.max())
    assert float((m - sym).abs().max()) > 1e-3 * float(per.max())
    a, _ = O.melspec_forward(x.numpy(), [5000, 4000, 3000], stft_conv=True)
    b, _ = O.melspec_forward(x.numpy(), [5000, 4000, 3000], stft_conv=False)
    assert a.shape == b.shape and float((a - b).abs().max()) > 1e-2


def test_ctc_collapse_rules():
    blank = 5
    assert O.ctc_collapse_ids([5, 5, 1, 1, 5, 1, 2, 2, 5], blank) == [1, 1, 2]
    assert O.ctc_collapse_ids([], blank) == []
    assert O.ctc_collapse_ids([0, 0, 0], blank) == [0]
    assert O.ctc_decode_strings(np.array([[0, 1, 1, 2, 0, 2]]), ["a", "b"]) == ["aba"]


def test_mel_filterbank_against_the_values_librosa_documents():
    """A4 (librosa.filters.mel, third party, absent here) has no fixture in the reference; the only published numbers are
    the two arrays printed in librosa's own documentation of filters.mel (cited from memory of the public docs, to the two
    significant digits they print): mel(sr=22050, n_fft=2048)[0, 1] = 0.016 and, with fmax=8000, 0.02; both first columns
    and the last columns are 0.  A weak pin -- it fixes the Slaney scale, the area normalisation and the bin grid, not the
    last bits -- and the restatement the library ships (frontend_tables) must equal the oracle's bit for bit."""
    from viet_asr_amd import frontend_tables as F
    m = np.asarray(O.slaney_mel_filterbank(sr=22050, n_fft=2048, n_mels=128))
    assert m.shape == (128, 1025) and m.dtype == np.float32
    assert round(float(m[0, 1]), 3) == 0.016 and m[0, 0] == 0 and m[0, -1] == 0 and m[-1, 0] == 0 and m[-1, -1] == 0
    m8 = np.asarray(O.slaney_mel_filterbank(sr=22050, n_fft=2048, n_mels=128, fmax=8000))
    assert round(float(m8[0, 1]), 2) == 0.02 and not m8[:, 800:].any()          # nothing above 8 kHz (bin 743)
    # Slaney area normalisation: every triangle integrates to ~1 over its band in Hz (2 / (f[i+2] - f[i]) peak scaling)
    hz_per_bin = 22050 / 2048
    assert np.allclose(m.sum(axis=1) * hz_per_bin, 1.0, atol=0.06)
    assert np.array_equal(np.asarray(F.mel_filterbank(sr=22050, n_fft=2048, n_mels=128)), m)
    assert np.array_equal(np.asarray(F.mel_filterbank()), np.asarray(O.slaney_mel_filterbank()))
