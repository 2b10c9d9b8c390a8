"""-m gpu: HIP path (through the C ABI) vs the golden fixtures of the real reference and vs the oracle."""
import os
import numpy as np
import pytest
import torch

from conftest import BAND_CASES, GOLDEN_CASES, OPTION_CASES, REAL_AUDIO_CASES, load_golden, load_real_audio_golden, oracle_pre_kwargs

pytestmark = pytest.mark.gpu

# Tolerances (fp32 path, stated per north_star): the HIP kernels accumulate in a different order
# than MKL-DNN / pocketfft, so values agree to fp32 round-off amplified through <= 86 layers;
# predictions and transcripts must be IDENTICAL (the smallest top-2 margin of a fixture is 9e-4; measured log-prob
# errors are <= 1.7e-4 on those fixtures: the bound is 3x that, it was 12x in round 1).
MEL_TOL = 2e-4
LOGP_TOL = 5e-4
# ... for log-probs up to ~100 in magnitude.  The synthetic 15x5 model is far peakier on long clips (|log-prob| up to
# 480 on the 10 s fixture, where one fp32 ulp is 3e-5): the bound follows the magnitude, 2e-5 relative.
LOGP_REL = 2e-5


def logp_tol(ref):
    return max(LOGP_TOL, LOGP_REL * float(np.abs(np.asarray(ref)).max()))


def _record(test, **kw):
    """Measured errors go to gpurun_out/parity_errors.jsonl (when that scratch directory exists): the tolerances above
    are set from these numbers, not guessed."""
    import json, os
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(d):
        with open(os.path.join(d, "parity_errors.jsonl"), "a") as f:
            f.write(json.dumps(dict(test=test, **{k: (float(v) if hasattr(v, "__float__") else v) for k, v in kw.items()})) + "\n")


def _engine(cfg, enc_sd, dec_sd, gemm=None):
    from viet_asr_amd.engine import QuartzNetCTC
    return QuartzNetCTC(cfg, enc_sd, dec_sd, gemm=gemm)


@pytest.mark.parametrize("gemm", ["f16x2", "bf16x3", "fp32"])
@pytest.mark.parametrize("name", GOLDEN_CASES + OPTION_CASES)
def test_fused_path_matches_reference_goldens(gpu, name, gemm):
    """The three fp32-equivalent GEMM arithmetics (2 x fp16 scaled split operands, 3 x bf16 split operands, exact-fp32
    MFMA) against the reference goldens, same tolerance for all."""
    g, cfg, sig, lens, enc_sd, dec_sd = load_golden(name)
    eng = _engine(cfg, enc_sd, dec_sd, gemm)
    r = eng.forward(torch.from_numpy(sig).to(gpu), torch.from_numpy(lens).to(gpu), want_logp=True)
    torch.cuda.synchronize()
    logp = r["logp"].cpu().numpy()
    assert logp.shape == g["logp"].shape
    assert np.isfinite(logp).all()
    err = np.abs(logp - g["logp"]).max()
    _record("goldens", name=name, gemm=gemm, err=err, scale=np.abs(g["logp"]).max(), min_margin=g["min_margin"])
    assert err <= logp_tol(g["logp"]), err
    assert (r["enc_len"].cpu().numpy() == g["enc_len"]).all()
    assert (r["pred"].cpu().numpy() == g["pred"]).all()
    assert eng.texts(r["ids"], r["id_len"]) == [str(s) for s in g["hyp"]]


# Band-limited audio (nothing above 4 kHz but the anti-alias filter's -85 dB floor).  Measured on the two fixtures (round 5,
# profiles/r05_parity_errors.jsonl): log-prob error 1.6e-4 ... 5.2e-4 at |log-prob| 46 ... 83, features 1.8e-5 / 3.6e-5 over
# ALL bins, 0 of 692 frames differ -- inside the ordinary goldens' tolerance, so they get no allowance.  (What DOES need
# one is a recording whose upper bins sit exactly at the log guard -- the real 8 kHz file below, the 512 x 30 s shard of
# test_gpu_flips.py -- where the reference's (x - mean) / (std + 1e-5) divides by a std of 1e-3)
BAND_LOGP_FACTOR = 1


@pytest.mark.parametrize("gemm", ["f16x2", "fp32"])
@pytest.mark.parametrize("name", BAND_CASES)
def test_band_limited_reference_goldens(gpu, name, gemm):
    """batched, ragged, band-limited audio (the reference's stated 8 kHz-sourced domain, README.md:21)
    through the imported reference (make_golden.py) against the device, end to end from the waveform: predictions and
    transcripts IDENTICAL on every frame (padded frames included), log-probs within BAND_LOGP_FACTOR x the goldens' bound,
    features of the bins that hold signal within MEL_TOL.  The vi case carries the shipped Vietnamese head (blank on every
    frame of an untrained encoder: log-probs compared) and a seeded head (a transcript that bites)."""
    from viet_asr_amd import synth
    g, cfg, sig, lens, enc_sd, dec_sd = load_golden(name)
    heads = [("head", dec_sd, "logp", "pred", "hyp", "margin")]
    if "logp_syn" in g.files:
        heads.append(("seeded", synth.decoder_state_dict(1024, len(cfg["labels"]) + 1, int(g["seed"])), "logp_syn", "pred_syn", "hyp_syn", "margin_syn"))
    wav, ln = torch.from_numpy(sig).to(gpu), torch.from_numpy(lens).to(gpu)
    for tag, head, k_logp, k_pred, k_hyp, k_margin in heads:
        eng = _engine(cfg, enc_sd, head, gemm)
        r = eng.forward(wav, ln, want_logp=True)
        torch.cuda.synchronize()
        logp, pred = r["logp"].cpu().numpy(), r["pred"].cpu().numpy()
        err = float(np.abs(logp - g[k_logp]).max())
        flips = pred != g[k_pred]
        _record("band_goldens", name=name, head=tag, gemm=gemm, err=err, scale=float(np.abs(g[k_logp]).max()), frames=int(pred.size),
                flips=int(flips.sum()), min_margin=float(g[k_margin].min()))
        assert err <= BAND_LOGP_FACTOR * logp_tol(g[k_logp]), (tag, err)
        assert (r["enc_len"].cpu().numpy() == g["enc_len"]).all()
        assert not flips.any(), (tag, int(flips.sum()), g[k_margin][flips].tolist())
        assert eng.texts(r["ids"], r["id_len"]) == [str(s) for s in g[k_hyp]]
    # the front end on its own: bins with signal (no filter weight above 4.2 kHz excluded) to the goldens' tolerance
    from viet_asr_amd import stages
    mel, seq = stages.melspec(eng.handle, wav, ln)
    mel = mel.cpu().numpy()
    assert (seq.cpu().numpy() == g["seq"]).all() and mel.shape == g["mel"].shape
    fb = g["fb"]
    lower = np.array([fb[m, int(4200 / 8000 * 256):].sum() == 0 for m in range(64)])
    e_low = float(np.abs(mel[:, lower] - g["mel"][:, lower]).max())
    e_all = float(np.abs(mel - g["mel"]).max())
    _record("band_goldens_mel", name=name, err_bins_below_4k=e_low, err_all_bins=e_all, bins_below=int(lower.sum()))
    assert e_low <= MEL_TOL and e_all <= MEL_TOL, (e_low, e_all)


def _oracle(cfg, sig, lens, enc_sd, dec_sd):
    from oracle import quartznet_oracle as O
    return O.forward_all(sig, lens, enc_sd, dec_sd, cfg["JasperEncoder"]["jasper"])


@pytest.mark.parametrize("name", ["vi12x1_b3_ragged", "en15x5_b2_ragged"])
def test_neural_module_dag_matches_goldens(gpu, name):
    """The reference's own wiring (infer.py:146-160 with the greedy decoder) through our factory."""
    from viet_asr_amd import asr as nemo_asr
    from viet_asr_amd.core import DeviceType, NeuralModuleFactory
    from viet_asr_amd.helpers import post_process_predictions
    g, cfg, sig, lens, enc_sd, dec_sd = load_golden(name)
    nf = NeuralModuleFactory(placement=DeviceType.GPU)
    dl = nemo_asr.AudioDataLayer(sample_rate=16000)
    pre = nemo_asr.AudioToMelSpectrogramPreprocessor(**dict(cfg["AudioToMelSpectrogramPreprocessor"], dither=0, pad_to=0))
    enc = nemo_asr.JasperEncoder(feat_in=64, **cfg["JasperEncoder"])
    dec = nemo_asr.JasperDecoderForCTC(feat_in=1024, num_classes=len(cfg["labels"]))
    greedy = nemo_asr.GreedyCTCDecoder()
    enc.load_state_dict({k: torch.as_tensor(v) for k, v in enc_sd.items()})
    dec.load_state_dict({k: torch.as_tensor(v) for k, v in dec_sd.items()})
    a, al = dl()
    mel, ml = pre(input_signal=a, length=al)
    e, el = enc(audio_signal=mel, length=ml)
    lp = dec(encoder_output=e)
    pred = greedy(log_probs=lp)
    dl.set_batch([sig[b, : lens[b]] for b in range(len(lens))])
    out = nf.infer(tensors=[mel, ml, el, lp, pred], verbose=False)
    mel_v, ml_v, el_v, lp_v, pred_v = [o[0] for o in out]
    assert np.abs(mel_v.numpy() - g["mel"]).max() <= MEL_TOL
    assert ml_v.dtype == torch.int64 and (ml_v.numpy() == g["seq"]).all()
    assert el_v.dtype == torch.float32 and (el_v.numpy() == g["enc_len"]).all()          # quirk Q3
    assert np.abs(lp_v.numpy() - g["logp"]).max() <= LOGP_TOL
    assert pred_v.dtype == torch.int64 and (pred_v.numpy() == g["pred"]).all()
    assert post_process_predictions([pred_v], cfg["labels"]) == [str(s) for s in g["hyp"]]


@pytest.mark.parametrize("name", OPTION_CASES)
def test_front_end_options_no_shipped_config_uses(gpu, name):
    """log_zero_guard_type="clamp" (parts/features.py:272-273) and normalize="all_features" (:31-39): fixtures generated by the
    imported reference with those constructor arguments; the NeuralModule built from the same kwargs (the reference's
    `AudioToMelSpectrogramPreprocessor(**cfg)`, infer.py:102-103) against the reference's features, frame by frame, and
    against the oracle with the same options."""
    from viet_asr_amd import asr as nemo_asr
    from viet_asr_amd.core import DeviceType, NeuralModuleFactory
    from oracle import quartznet_oracle as O
    g, cfg, sig, lens, enc_sd, dec_sd = load_golden(name)
    NeuralModuleFactory(placement=DeviceType.GPU)
    pre_cfg = dict(cfg["AudioToMelSpectrogramPreprocessor"], dither=0, pad_to=0)
    pre_cfg.pop("feat_type", None)
    pre = nemo_asr.AudioToMelSpectrogramPreprocessor(**pre_cfg)
    mel, seq = pre(force_pt=True, input_signal=torch.from_numpy(sig).to(gpu), length=torch.from_numpy(lens).to(gpu))
    assert (seq.cpu().numpy() == g["seq"]).all()
    err = float(np.abs(mel.cpu().numpy() - g["mel"]).max())
    ref_mel, _ = O.melspec_forward(sig, lens, **oracle_pre_kwargs(cfg))
    _record("front_end_options", name=name, err=err, err_vs_oracle=float((mel.cpu() - ref_mel).abs().max()))
    assert err <= MEL_TOL, err
    assert (mel.cpu().numpy()[g["mel"] == 0] == 0).all()
    with pytest.raises(ValueError):
        nemo_asr.AudioToMelSpectrogramPreprocessor(**dict(pre_cfg, log_zero_guard_type="floor"))


def test_dither_is_the_reference_torch_call_on_the_device(gpu):
    """dither (parts/features.py:250-251: `x += dither * torch.randn_like(x)`): the module makes the SAME torch call on the same
    device tensor, in place like the reference, so under one torch.manual_seed it is bit-for-bit the dither = 0 module fed with
    signal + noise; the features of that dithered signal against the oracle.  (The reference-generated fixture of this case
    draws from the CPU generator: tests/test_oracle_golden.py replays it with the oracle.)"""
    from viet_asr_amd import asr as nemo_asr
    from viet_asr_amd.core import DeviceType, NeuralModuleFactory
    from oracle import quartznet_oracle as O
    g, cfg, sig, lens, enc_sd, dec_sd = load_golden("vi12x1_b2_dither_1e3")
    NeuralModuleFactory(placement=DeviceType.GPU)
    pre_cfg = dict(cfg["AudioToMelSpectrogramPreprocessor"], pad_to=0)
    pre_cfg.pop("feat_type", None)
    assert pre_cfg["dither"] == 1e-3
    pre_d = nemo_asr.AudioToMelSpectrogramPreprocessor(**pre_cfg)
    pre_0 = nemo_asr.AudioToMelSpectrogramPreprocessor(**dict(pre_cfg, dither=0))
    x, n = torch.from_numpy(sig).to(gpu), torch.from_numpy(lens).to(gpu)
    torch.manual_seed(7)
    xa = x.clone()
    mel_d, seq = pre_d(force_pt=True, input_signal=xa, length=n)
    torch.manual_seed(7)
    xb = x.clone()
    xb += 1e-3 * torch.randn_like(xb)
    mel_0, _ = pre_0(force_pt=True, input_signal=xb, length=n)
    assert torch.equal(xa, xb) and not torch.equal(xa, x)          # in place, like the reference
    assert torch.equal(mel_d, mel_0) and (seq.cpu().numpy() == g["seq"]).all()
    assert 0.9e-3 < float((xa - x).std()) < 1.1e-3
    ref_mel, _ = O.melspec_forward(xa.cpu().numpy(), lens)
    err = float((mel_d.cpu() - ref_mel).abs().max())
    _record("front_end_dither", err_vs_oracle=err)
    assert err <= MEL_TOL, err


def test_stage_entry_points_against_oracle(gpu):
    """Each C-ABI stage fed with the ORACLE's input for that stage (errors do not compound)."""
    from viet_asr_amd import _lib, configs, stages, synth
    from viet_asr_amd.engine import blocks_from_config
    from viet_asr_amd.frontend_tables import frontend_description
    from oracle import quartznet_oracle as O
    cfg = configs.builtin("quartznet12x1_vi")
    jas = cfg["JasperEncoder"]["jasper"]
    enc_sd, dec_sd = synth.encoder_state_dict(jas, 64, 21), synth.decoder_state_dict(1024, 91, 21)
    sig, lens = synth.audio_batch(5, 48000, 21, ragged=True)
    lens[1] = 160 * 199              # Q2: exact multiple of the hop
    sig[1, lens[1]:] = 0
    ref = _oracle(cfg, sig, lens, enc_sd, dec_sd)
    hp = _lib.Handle(frontend=frontend_description(cfg["AudioToMelSpectrogramPreprocessor"])); hp.finalize()
    he = _lib.Handle(feat_in=64, blocks=blocks_from_config(jas)); he.load_state_dict(enc_sd); he.finalize()
    hd = _lib.Handle(dec_feat_in=1024, num_classes=91); hd.load_state_dict(dec_sd); hd.finalize()
    mel, seq = stages.melspec(hp, torch.from_numpy(sig).to(gpu), torch.from_numpy(lens).to(gpu))
    assert (seq.cpu() == ref["seq"]).all()
    assert (mel.cpu() - ref["mel"]).abs().max() <= MEL_TOL
    assert (mel.cpu()[ref["mel"] == 0] == 0).all()                   # masked frames are exactly pad_value
    enc, elen = stages.encoder(he, ref["mel"].to(gpu), ref["seq"].to(gpu), 1024)
    assert (elen.cpu() == ref["enc_len"]).all()
    scale = float(ref["enc"].abs().max())
    assert (enc.cpu() - ref["enc"]).abs().max() <= 2e-5 * max(scale, 1.0)
    logp = stages.decoder(hd, ref["enc"].to(gpu))
    assert (logp.cpu() - ref["logp"]).abs().max() <= 2e-4
    assert (stages.greedy_argmax(ref["logp"].to(gpu)).cpu() == ref["pred"]).all()
    ids, n = stages.ctc_collapse(ref["pred"].to(gpu), 90)
    for b in range(5):
        assert ids[b, : n[b]].cpu().tolist() == O.ctc_collapse_ids(ref["pred"][b].numpy(), 90)


def test_edge_cases(gpu):
    from viet_asr_amd import configs, stages, synth
    cfg = configs.builtin("quartznet12x1_vi")
    jas = cfg["JasperEncoder"]["jasper"]
    enc_sd, dec_sd = synth.encoder_state_dict(jas, 64, 5), synth.decoder_state_dict(1024, 91, 5)
    eng = _engine(cfg, enc_sd, dec_sd)
    # shortest input torch.stft accepts (reflect pad needs > n_fft/2 samples), one very short row in a batch
    sig, lens = synth.audio_batch(3, 2000, 5, ragged=False)
    lens[:] = [2000, 257, 1]
    for b in range(3):
        sig[b, lens[b]:] = 0
    ref = _oracle(cfg, sig, lens, enc_sd, dec_sd)
    r = eng.forward(torch.from_numpy(sig).to(gpu), torch.from_numpy(lens).to(gpu), want_logp=True)
    lp = r["logp"].cpu()
    # a row with seq_len == 1 has NaN statistics in the reference (unbiased std of one frame): same here
    assert torch.isnan(ref["logp"][2]).all() == torch.isnan(lp[2]).all()
    ok = ~torch.isnan(ref["logp"])
    assert (lp[ok] - ref["logp"][ok]).abs().max() <= logp_tol(ref["logp"][ok])
    assert (r["pred"].cpu()[:2] == ref["pred"][:2]).all()
    with pytest.raises(ValueError):          # torch.stft refuses reflect padding of <= n_fft/2 samples
        eng.forward(torch.zeros(1, 256, device=gpu), torch.tensor([256], device=gpu))
    # all-blank and all-repeat rows through the collapse
    pred = torch.tensor([[90] * 7, [3] * 7, [3, 90, 3, 3, 90, 90, 4]], device=gpu)
    ids, n = stages.ctc_collapse(pred, 90)
    assert n.tolist() == [0, 1, 3] and ids[2, :3].tolist() == [3, 3, 4]
    # argmax ties: lowest index wins (quirk Q6)
    lp0 = torch.zeros(1, 2, 29, device=gpu)
    lp0[0, 1, 5] = lp0[0, 1, 9] = 1.0
    assert stages.greedy_argmax(lp0).tolist() == [[0, 5]]


def test_full_size_properties(gpu):
    """BASELINE.json config 2 size (12x1_vi, B=32 x 10 s): properties that need no oracle run."""
    from viet_asr_amd import configs, stages, synth
    cfg = configs.builtin("quartznet12x1_vi")
    jas = cfg["JasperEncoder"]["jasper"]
    enc_sd, dec_sd = synth.encoder_state_dict(jas, 64, 2), synth.decoder_state_dict(1024, 91, 2)
    eng = _engine(cfg, enc_sd, dec_sd)
    sig, lens = synth.audio_batch(32, 160000, 2, ragged=True)
    wav, ln = torch.from_numpy(sig).to(gpu), torch.from_numpy(lens).to(gpu)
    r1 = eng.forward(wav, ln, want_logp=True)
    r2 = eng.forward(wav, ln, want_logp=True)
    assert torch.equal(r1["pred"], r2["pred"]) and torch.equal(r1["logp"], r2["logp"])     # deterministic
    assert r1["logp"].shape == (32, 501, 91)
    assert (torch.logsumexp(r1["logp"], -1).abs().max() < 1e-4)                            # rows are log-distributions
    assert torch.equal(r1["logp"].argmax(-1), r1["pred"])
    assert r1["enc_len"].tolist() == [float((int(np.ceil(l / 160)) - 1) // 2 + 1) for l in lens]
    # batch independence up to the padded-row reflect quirk (Q5): full-length rows do not depend on the others.  (Within the
    # parity tolerance, not to the bit: since round 4 a batch of this size runs its 256-channel sub-blocks through the fused
    # kernel's 64-frame form and a lone utterance does not -- the two round differently, measured 2e-5 at |log-prob| 40;
    # bit-identical rows whatever the batch is what vasr_set_row_independent promises, test_row_independent_*.)
    full = [b for b in range(32) if lens[b] == 160000][:1]
    solo = eng.forward(wav[full], ln[full], want_logp=True)
    assert (solo["logp"][0] - r1["logp"][full[0]]).abs().max() <= logp_tol(solo["logp"][0].cpu()) / 4
    # collapse is idempotent on its own output re-expanded with blanks
    ids, n = r1["ids"], r1["id_len"]
    row = ids[0, : n[0]].long()
    expanded = torch.stack([row, torch.full_like(row, 90)], 1).reshape(1, -1)
    ids2, n2 = stages.ctc_collapse(expanded, 90)
    assert ids2[0, : n2[0]].tolist() == row.tolist()


def test_vietasr_class_end_to_end(gpu, tmp_path):
    """infer.py's VietASR on synthetic checkpoints written in the reference's state_dict format: DAG path (greedy and
    beam wiring), fused batch path, 8 kHz input resampled on the device."""
    from viet_asr_amd import configs, synth
    from viet_asr_amd.infer import VietASR
    from oracle import audio_oracle as AO
    from oracle import quartznet_oracle as O
    cfg = configs.builtin("quartznet12x1_vi")
    jas = cfg["JasperEncoder"]["jasper"]
    enc_sd, dec_sd = synth.encoder_state_dict(jas, 64, 8), synth.decoder_state_dict(1024, 91, 8)
    enc_p, dec_p = str(tmp_path / "JasperEncoder-STEP-1.pt"), str(tmp_path / "JasperDecoderForCTC-STEP-1.pt")
    torch.save({k: torch.as_tensor(v) for k, v in enc_sd.items()}, enc_p)
    torch.save({k: torch.as_tensor(v) for k, v in dec_sd.items()}, dec_p)
    asr = VietASR("quartznet12x1_vi", enc_p, dec_p, device="gpu", decoder="greedy")
    sig, lens = synth.audio_batch(3, 24000, 8, ragged=True)
    utts = [sig[b, : lens[b]] for b in range(3)]
    for b in range(3):                     # one utterance at a time, like infer.py
        ref = O.forward_all(utts[b][None], np.array([len(utts[b])]), enc_sd, dec_sd, jas)
        assert asr.transcribe(utts[b]) == O.ctc_decode_strings(ref["pred"], cfg["labels"])[0]
    ref = O.forward_all(sig, lens, enc_sd, dec_sd, jas)            # padded batch (quirks Q4/Q5 apply)
    assert asr.transcribe_batch(utts) == O.ctc_decode_strings(ref["pred"], cfg["labels"])
    # 8 kHz int16 input: scaled + resampled on the device, then the same path
    x8 = (synth.audio_batch(1, 12000, 9)[0][0] * 32767).astype(np.int16)
    up = AO.resample(x8.astype(np.float32) / 32768.0, 8000, 16000)
    ref8 = O.forward_all(up[None], np.array([len(up)]), enc_sd, dec_sd, jas)
    assert asr.transcribe(x8, sample_rate=8000) == O.ctc_decode_strings(ref8["pred"], cfg["labels"])[0]
    beam = VietASR("quartznet12x1_vi", enc_p, dec_p, device="gpu", decoder="beam", beam_width=8, lm_path=None)
    out = beam.transcribe(utts[0])
    assert isinstance(out, str)
    # batched beam search, row-independent: the transcripts of the reference-style one-at-a-time calls; and the greedy
    # batch in the same mode equals the one-at-a-time greedy calls (the padded batch above does not, quirks Q4/Q5)
    assert beam.transcribe_batch(utts, decoder="beam", row_independent=True) == [beam.transcribe(u) for u in utts]
    assert asr.transcribe_batch(utts, row_independent=True) == [asr.transcribe(u) for u in utts]
    with pytest.raises(ValueError):
        asr.transcribe_batch(utts, decoder="beam")
    with pytest.raises(AssertionError):
        VietASR("quartznet12x1_vi", str(tmp_path / "missing.pt"), dec_p)


def test_module_by_module_path_equals_the_fused_row_independent_path(gpu):
    """VietASR.transcribe (infer.py's DAG of NeuralModules, one utterance per call) against transcribe_batch(row_independent=True)
    (the fused one-call path on a batch), which promises per row "what transcribe returns for that signal alone": sixty cases of
    tests/devtools/fuzz_dag.py -- 0.2-12 s, three levels, float and int16, 16 and 8 kHz, both shipped model families.  Round 6: the
    per-module CTC head ran 3 x bf16 where the fused path runs 2 x fp16 (same tolerance, other bits): 38 of 181 272 signals came
    out a character different; vasr_decoder_logsoftmax_f32 now takes the port tensor's maxima and runs the fused path's arithmetic
    (24 346 cases / 167 289 signals, 0 differences: profiles/r06_fuzz_campaign.txt)."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "devtools"))
    import fuzz_dag
    bad = [m for m in (fuzz_dag.dag_case(c) for c in range(60)) if m]
    assert not bad, bad
    assert fuzz_dag.STATS["signals"] > 100 and fuzz_dag.STATS["chars"] > 1000


@pytest.mark.parametrize("cin,kind", [(256, "randn"), (512, "randn"), (1024, "randn"), (512, "relu"), (512, "wide")])
def test_split_gemms_are_as_accurate_as_fp32_mfma(gpu, cin, kind):
    """Isolated 1x1-conv GEMM (cin -> 512 channels) against an fp64 reference: neither split arithmetic (3 x bf16, six
    products; 2 x fp16 scaled, three products) may be less accurate than the exact-fp32 MFMA chain, and all three must
    agree to fp32 round-off.  Inputs: Gaussian; ReLU'd Gaussian (what the layers actually see); and "wide" -- rows whose
    magnitudes span 2^-24 ... 2^6 inside one utterance, far more than the 18 octaves the fp16 split carries at full
    precision, where its error must stay below fp32 round-off of the LARGEST terms (absolute, not relative)."""
    from viet_asr_amd import _lib
    import ctypes as C
    L = _lib.dev_lib()      # include/vasr_devtools.h lives in the devtools build
    B, T, cout = 2, 300, 512
    ld = int(L.vasr_padded_frames(T))
    g = torch.Generator().manual_seed(cin + len(kind))
    x = torch.randn(B, cin, ld, generator=g)
    if kind == "relu":
        x = torch.relu(x)
    if kind == "wide":
        x = x * torch.exp2(torch.randint(-24, 7, (B, cin, ld), generator=g).float())
    x[1] *= 37.5                                        # utterances of one batch at different scales
    x = x.to(gpu)
    w = (torch.randn(cout, cin, generator=g) / cin ** 0.5).contiguous()
    sc, sh = torch.ones(cout, device=gpu), torch.zeros(cout, device=gpu)
    st = torch.cuda.current_stream().cuda_stream
    pk = torch.empty(cout * cin)
    _lib.check(L.vasr_pack_pointwise(w.data_ptr(), cout, cin, cout, pk.data_ptr()), L)
    pk3 = torch.empty(cout * cin * 3, dtype=torch.int16)
    _lib.check(L.vasr_pack_pointwise_bf16x3(w.data_ptr(), cout, cin, cout, pk3.data_ptr()), L)
    pk16 = torch.empty(cout * cin * 2, dtype=torch.int16)
    inv = C.c_float()
    _lib.check(L.vasr_pack_pointwise_f16x2(w.data_ptr(), cout, cin, cout, pk16.data_ptr(), C.byref(inv)), L)
    y32, y3, y16 = (torch.empty(B, cout, ld, device=gpu) for _ in range(3))
    amax = torch.full((2, B, 256), -1, dtype=torch.int32, device=gpu)       # per-wavefront maxima tables (x, y)
    w32, w3, w16 = pk.to(gpu), pk3.to(gpu), pk16.to(gpu)
    _lib.check(L.vasr_bench_pointwise(x.data_ptr(), w32.data_ptr(), sc.data_ptr(), sh.data_ptr(), B, cin, cout, T, y32.data_ptr(), st), L)
    _lib.check(L.vasr_bench_pointwise_bf16x3(x.data_ptr(), w3.data_ptr(), sc.data_ptr(), sh.data_ptr(), B, cin, cout, T, y3.data_ptr(), st), L)
    _lib.check(L.vasr_bench_pointwise_f16x2(x.data_ptr(), w16.data_ptr(), inv.value, sc.data_ptr(), sh.data_ptr(), B, cin, cout, T,
                                            y16.data_ptr(), amax.data_ptr(), 256, st), L)
    ref = torch.relu(torch.einsum("mk,bkt->bmt", w.double().to(gpu), x[:, :, :T].double()))
    err = lambda y: [float((y[b, :, :T].double() - ref[b]).abs().max()) for b in range(B)]
    e32, e3, e16 = err(y32), err(y3), err(y16)
    for b in range(B):          # per utterance: each has its own scale
        assert e3[b] <= 1.5 * e32[b] + 1e-7 * float(ref[b].max()), (cin, kind, b, e32, e3)
        assert e16[b] <= 1.5 * e32[b] + 1e-7 * float(ref[b].max()), (cin, kind, b, e32, e16)
        assert e32[b] <= 2e-6 * max(1.0, float(ref[b].max())), (e32, float(ref[b].max()))
    # the maxima the kernels publish: row 0 = max |x| over the valid frames, row 1 = max |y| (what the next layer's split uses)
    got = amax.view(torch.float32).amax(-1).cpu()
    assert torch.equal(got[0], x[:, :, :T].abs().amax((1, 2)).cpu())
    assert torch.equal(got[1], y16[:, :, :T].abs().amax((1, 2)).cpu())


def test_long_clips_config5_shape(gpu):
    """BASELINE config 5 geometry (30 s clips -> L=480000, T=3001, T'=1501) at a small batch, against the oracle."""
    from viet_asr_amd import configs, synth
    cfg = configs.builtin("quartznet15x5")
    jas = cfg["JasperEncoder"]["jasper"]
    enc_sd, dec_sd = synth.encoder_state_dict(jas, 64, 6), synth.decoder_state_dict(1024, 29, 6)
    eng = _engine(cfg, enc_sd, dec_sd)
    sig, lens = synth.audio_batch(2, 480000, 6, ragged=True)
    r = eng.forward(torch.from_numpy(sig).to(gpu), torch.from_numpy(lens).to(gpu), want_logp=True)
    ref = _oracle(cfg, sig, lens, enc_sd, dec_sd)
    assert r["logp"].shape == (2, 1501, 29)
    # logits span +-170 on this model (very peaky posteriors), so fp32 round-off is judged against that scale:
    # absolute tolerance + 5e-5 of the largest |log-prob| (measured: 4.7e-3 at scale 167 = 2.8e-5 relative)
    _record("long_clips", err=(r["logp"].cpu() - ref["logp"]).abs().max(), scale=ref["logp"].abs().max())
    assert (r["logp"].cpu() - ref["logp"]).abs().max() <= LOGP_TOL + 5e-5 * float(ref["logp"].abs().max())
    top2 = torch.topk(ref["logp"], 2, dim=-1).values
    # a flip needs the two leading classes inside twice the measured error (4.7e-3 at this scale): none elsewhere
    safe = (top2[..., 0] - top2[..., 1]) > 1e-2
    assert (r["pred"].cpu()[safe] == ref["pred"][safe]).all()
    assert float(safe.float().mean()) > 0.995


_ALT_PATH_SNIPPET = r"""
import sys, numpy as np, torch
sys.path.insert(0, {tests!r}); sys.path.insert(0, {root!r})
from conftest import load_golden
from viet_asr_amd.engine import QuartzNetCTC
bad = []
for name in ("vi12x1_b3_ragged", "en15x5_b2_ragged"):
    g, cfg, sig, lens, enc_sd, dec_sd = load_golden(name)
    eng = QuartzNetCTC(cfg, enc_sd, dec_sd)
    r = eng.forward(torch.from_numpy(sig).cuda(), torch.from_numpy(lens).cuda(), want_logp=True)
    torch.cuda.synchronize()
    err = float(np.abs(r["logp"].cpu().numpy() - g["logp"]).max())
    same = bool((r["pred"].cpu().numpy() == g["pred"]).all()) and eng.texts(r["ids"], r["id_len"]) == [str(s) for s in g["hyp"]]
    if err > 5e-4 or not same:
        bad.append((name, err, same))
print("ALT_PATH_OK" if not bad else "ALT_PATH_BAD %r" % bad)
"""


@pytest.mark.parametrize("env", [{"VASR_DW_PAIR": "0"}, {"VASR_NO_FUSED_RESIDUAL": "1"}, {"VASR_PW3_TILE": "3"},
                                 {"VASR_PW3_TILE": "4"}, {"VASR_SLICES": "2"},
                                 {"VASR_GEMM": "f16x2", "VASR_DW_PAIR": "0"}, {"VASR_GEMM": "f16x2", "VASR_NO_FUSED_RESIDUAL": "1"},
                                 {"VASR_GEMM": "f16x2", "VASR_PW3_TILE": "2"}, {"VASR_GEMM": "f16x2", "VASR_PW3_TILE": "3"},
                                 {"VASR_GEMM": "bf16x3", "VASR_PW3_TILE": "4"}, {"VASR_DW_MFMA": "0"}, {"VASR_DW_UPW": "3"},
                                 {"VASR_FUSED_MIN_TILES": "1"}, {"VASR_FUSED": "0"},
                                 {"VASR_FUSED_MIN_TILES": "1", "VASR_FUSED_TILE": "64"},
                                 {"VASR_FUSED_MIN_TILES": "1", "VASR_NO_FUSED_RESIDUAL": "1"}],
                         ids=lambda e: ",".join(f"{k}={v}" for k, v in e.items()))
def test_alternate_kernel_paths_match_goldens(gpu, env):
    """The kernels a default run does not pick (one-row depthwise, packed-FMA depthwise under the fp16-split GEMMs,
    two-GEMM residual, latency GEMM tiles, batch slices on side streams, the fused depthwise + pointwise kernel forced
    onto small batches or switched off) are selected by environment variables that only the DEVTOOLS build of the
    library reads (libvasr_hip_dev.so; the product library ignores them), once per process: each runs in a child
    process on that build."""
    import os
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    code = _ALT_PATH_SNIPPET.format(tests=here, root=os.path.dirname(here))
    from viet_asr_amd import _lib
    dev = os.path.join(os.path.dirname(_lib.LIB_PATH), "libvasr_hip_dev.so")
    out = subprocess.run([sys.executable, "-c", code], env={**os.environ, "VASR_LIB_PATH": dev, **env}, capture_output=True,
                         text=True, timeout=600)
    assert "ALT_PATH_OK" in out.stdout, (out.stdout[-2000:], out.stderr[-2000:])


@pytest.mark.parametrize("gemm", ["f16x2", "bf16x3"])
def test_results_do_not_depend_on_batch_size_or_tile_shape(gpu, gemm):
    """An utterance gives bit-identical log-probs alone (64x32 GEMM tiles, self-paired depthwise) and inside an odd
    batch of 67 equal-length clips (256x128 / 512x128 tiles, utterance pairs): every reduction of the two-kernel
    sub-blocks runs in the same order whatever the launch shape.  (The batch is sized so that neither call takes the fused
    depthwise + pointwise kernel -- 67 x 3 s = 134 tiles of 128 frames, 268 of 64: between its two fill rules; that kernel
    derives the split scale from a bound and rounds differently, within the parity tolerance.  Bit-identical rows for
    EVERY batch is what row-independent mode promises -- it never fuses -- and what the serving queue runs on:
    test_row_independent_batches_equal_unbatched_calls_bit_for_bit.)"""
    from viet_asr_amd import configs, synth
    cfg = configs.builtin("quartznet15x5")
    jas = cfg["JasperEncoder"]["jasper"]
    eng = _engine(cfg, synth.encoder_state_dict(jas, 64, 9), synth.decoder_state_dict(1024, 29, 9), gemm)
    sig, lens = synth.audio_batch(67, 48000, 9)
    sig[5] *= 1e-3                                     # rows at very different levels: the fp16 split scales per utterance
    sig[66] *= 40.0
    r = eng.forward(torch.from_numpy(sig).to(gpu), torch.from_numpy(lens).to(gpu), want_logp=True)
    for b in (0, 5, 33, 66):
        r1 = eng.forward(torch.from_numpy(sig[b:b + 1]).to(gpu), torch.from_numpy(lens[b:b + 1]).to(gpu), want_logp=True)
        assert torch.equal(r["logp"][b], r1["logp"][0])
        assert torch.equal(r["pred"][b], r1["pred"][0])


def test_default_mode_rows_across_batch_sizes_through_the_fused_kernel(gpu):
    """in the DEFAULT mode a row's log-probs depend (slightly) on the batch it sits in, because the 256-channel
    sub-blocks take the fused depthwise + pointwise kernel for some batch shapes (10 s clips: 64-frame tiles at 12-32
    utterances, 128-frame tiles from ~52) and two kernels for the others, and the fused form derives its fp16 split scale
    from a bound instead of the measured maximum.  This pins HOW MUCH: one 10 s clip alone (two kernels, latency GEMM),
    inside a batch of 16 (64-frame fused form) and inside a batch of 64 (128-frame form) -- identical predictions, log-probs
    within HALF the parity tolerance (measured, round 5: 2.7e-3 at |log-prob| 385, where the tolerance is 7.7e-3 and one
    float32 ulp 3e-5).  Bit-identical rows for EVERY batch shape
    are what vasr_set_row_independent promises (it never fuses), not the default mode: include/vasr.h says so."""
    from viet_asr_amd import configs, synth
    cfg = configs.builtin("quartznet15x5")
    jas = cfg["JasperEncoder"]["jasper"]
    eng = _engine(cfg, synth.encoder_state_dict(jas, 64, 21), synth.decoder_state_dict(1024, 29, 21))
    sig, lens = synth.audio_batch(64, 160000, 21)
    wav, ln = torch.from_numpy(sig).to(gpu), torch.from_numpy(lens).to(gpu)
    eng.handle.profile_begin()
    r1 = eng.forward(wav[:1], ln[:1], want_logp=True)
    torch.cuda.synchronize()
    assert eng.handle.profile_end()["fused"]["launches"] == 0
    tol = logp_tol(r1["logp"].cpu().numpy()) / 2
    for B in (16, 64):
        eng.handle.profile_begin()
        rb = eng.forward(wav[:B], ln[:B], want_logp=True)
        torch.cuda.synchronize()
        assert eng.handle.profile_end()["fused"]["launches"] == 30          # every 256-channel sub-block went through the fused kernel
        err = float((rb["logp"][0] - r1["logp"][0]).abs().max())
        _record("default_mode_batch_dependence", batch=B, err=err, scale=float(r1["logp"].abs().max()), tol=tol)
        assert err <= tol, (B, err, tol)
        assert torch.equal(rb["pred"][0], r1["pred"][0])


@pytest.mark.parametrize("name", GOLDEN_CASES)
def test_reduced_bf16x2_mode_is_opt_in_and_stays_inside_the_tolerance(gpu, name):
    """vasr_set_gemm_mode(h, 2): 16-bit operand significands, three cross terms.  Not the default and not a parity
    claim -- this pins what the mode costs on the reference fixtures: identical predictions, log-prob error 5-10x the
    default mode's (measured 9e-6 ... 1.5e-3 on the short fixtures, 3.4e-2 at |log-prob| 480 on the 10 s one): bounded by
    ten times the default mode's tolerance."""
    g, cfg, sig, lens, enc_sd, dec_sd = load_golden(name)
    eng = _engine(cfg, enc_sd, dec_sd, "bf16x2")
    r = eng.forward(torch.from_numpy(sig).to(gpu), torch.from_numpy(lens).to(gpu), want_logp=True)
    assert np.abs(r["logp"].cpu().numpy() - g["logp"]).max() <= 10 * logp_tol(g["logp"])
    assert (r["pred"].cpu().numpy() == g["pred"]).all()
    default = _engine(cfg, enc_sd, dec_sd)          # the library default is NOT this mode
    rd = default.forward(torch.from_numpy(sig).to(gpu), torch.from_numpy(lens).to(gpu), want_logp=True)
    assert np.abs(rd["logp"].cpu().numpy() - g["logp"]).max() <= np.abs(r["logp"].cpu().numpy() - g["logp"]).max() + 1e-6


def _random_architecture(rng):
    """A JasperEncoder block list inside what the library accepts (filters % 128 == 0, dense convs 1x1 only, stride
    only on a residual-free block), with kernel sizes, repeats, dilation and residuals the builtin models do not use."""
    def blk(filters, kernel, repeat, stride=1, dilation=1, residual=False, separable=True):
        return dict(filters=int(filters), repeat=int(repeat), kernel=[int(kernel)], stride=[int(stride)],
                    dilation=[int(dilation)], dropout=0.0, residual=bool(residual), separable=bool(separable))
    odd = lambda lo, hi: int(rng.integers(lo // 2, hi // 2 + 1)) * 2 + 1
    blocks = [blk(rng.choice([128, 256]), odd(3, 41), rng.integers(1, 3), stride=rng.choice([1, 2]))]
    for _ in range(int(rng.integers(1, 4))):
        blocks.append(blk(rng.choice([128, 256, 384, 512]), odd(3, 99), rng.integers(1, 4), residual=rng.random() < 0.6))
    blocks.append(blk(rng.choice([128, 256]), odd(3, 91), 1, dilation=2))
    blocks.append(blk(rng.choice([128, 384]), 1, 1, separable=False))
    return blocks


@pytest.mark.parametrize("gemm", ["f16x2", "bf16x3"])
@pytest.mark.parametrize("seed", range(8))
def test_random_architectures_and_shapes_match_oracle(gpu, seed, gemm):
    import copy
    from viet_asr_amd import configs, synth
    rng = np.random.default_rng(1000 + seed)
    cfg = copy.deepcopy(configs.builtin("quartznet15x5"))
    jas = cfg["JasperEncoder"]["jasper"] = _random_architecture(rng)
    enc_sd = synth.encoder_state_dict(jas, 64, seed)
    dec_sd = synth.decoder_state_dict(jas[-1]["filters"], len(cfg["labels"]) + 1, seed)
    eng = _engine(cfg, enc_sd, dec_sd, gemm)
    B, L = int(rng.integers(1, 6)), int(rng.integers(1500, 30000))
    sig, lens = synth.audio_batch(B, L, seed, ragged=True)
    lens[int(rng.integers(0, B))] = L                        # the collate pads to the longest row
    lens[int(rng.integers(0, B))] = max(300, lens.min() // 3)
    for b in range(B):
        sig[b, lens[b]:] = 0
    ref = _oracle(cfg, sig, lens, enc_sd, dec_sd)
    r = eng.forward(torch.from_numpy(sig).to(gpu), torch.from_numpy(lens).to(gpu), want_logp=True)
    lp, want = r["logp"].cpu(), ref["logp"]
    assert lp.shape == want.shape, (jas, B, L)
    _record("random_arch", seed=seed, gemm=gemm, err=(lp - want).abs().max(), scale=want.abs().max())
    assert (lp - want).abs().max() <= logp_tol(want), (float((lp - want).abs().max()), jas, B, L)
    assert r["enc_len"].cpu().tolist() == ref["enc_len"].tolist()
    top2 = want.topk(2, -1).values
    clear = (top2[..., 0] - top2[..., 1]) > 2 * logp_tol(want)   # a flip needs both leading classes inside the tolerance
    assert (r["pred"].cpu()[clear] == ref["pred"][clear]).all()


def test_random_architectures_in_batches_of_every_size_class(gpu):
    """Thirty cases of tests/devtools/fuzz_encoder.py: the random block lists of the test above, half of them with a stack of
    256-channel K = 33 / 39 sub-blocks, in ragged batches of 1-5 / 6-20 / 21-72 rows of short clips and a randomly drawn GEMM
    arithmetic -- through the PRODUCT library's own kernel choice (the batch <= 5 latency GEMM, the 512 x 128 ... 128 x 64 tile
    rule, the fused kernel where it fills the chip).  Round-6 campaign: 2 133 such cases, worst error 0.09 x the tolerance
    (profiles/r06_fuzz_campaign.txt)."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "devtools"))
    import fuzz_encoder
    bad = [m for m in (fuzz_encoder.encoder_case(c) for c in range(30)) if m]
    assert not bad, bad
    _record("encoder_fuzz", **{k: (v if not isinstance(v, list) else str(v)) for k, v in fuzz_encoder.STATS.items()})
    assert min(fuzz_encoder.STATS["by_batch_class"]) > 0


def test_five_minute_signal_in_one_pass_matches_oracle(gpu):
    """The reference CLI skips files longer than 10 s (infer.py:201-203); here a long recording is one pass (T = 30 001
    frames, no halo tiling): same transcript as the oracle."""
    from viet_asr_amd import configs, synth
    from oracle import quartznet_oracle as O
    cfg = configs.builtin("quartznet12x1_vi")
    jas = cfg["JasperEncoder"]["jasper"]
    enc_sd, dec_sd = synth.encoder_state_dict(jas, 64, 4), synth.decoder_state_dict(1024, 91, 4)
    eng = _engine(cfg, enc_sd, dec_sd)
    n = 5 * 60 * 16000
    x = (0.1 * np.random.default_rng(5).standard_normal(n)).astype(np.float32)
    ref = O.forward_all(x[None], np.array([n]), enc_sd, dec_sd, jas)
    text = eng.transcribe([x])[0]
    assert len(text) > 1000 and text == O.ctc_decode_strings(ref["pred"], cfg["labels"])[0]


def test_strongly_ragged_batch_skipped_tiles_match_oracle(gpu):
    """Rows far shorter than the batch maximum: most time tiles of those rows lie past their length, where the GEMM
    skips its K loop and the depthwise kernel writes zeros -- the padded frames must still come out as the reference
    computes them (it decodes them, quirk Q4)."""
    from viet_asr_amd import configs, synth
    cfg = configs.builtin("quartznet12x1_vi")
    jas = cfg["JasperEncoder"]["jasper"]
    enc_sd, dec_sd = synth.encoder_state_dict(jas, 64, 9), synth.decoder_state_dict(1024, 91, 9)
    eng = _engine(cfg, enc_sd, dec_sd)
    L = 160000
    sig, lens = synth.audio_batch(5, L, 9, ragged=False)
    lens[:] = [L // 8, L, L // 3, 400, 41000]
    for b in range(5):
        sig[b, lens[b]:] = 0
    ref = _oracle(cfg, sig, lens, enc_sd, dec_sd)
    r = eng.forward(torch.from_numpy(sig).to(gpu), torch.from_numpy(lens).to(gpu), want_logp=True)
    _record("strongly_ragged", err=(r["logp"].cpu() - ref["logp"]).abs().max(), scale=ref["logp"].abs().max())
    assert (r["logp"].cpu() - ref["logp"]).abs().max() <= logp_tol(ref["logp"])
    assert r["enc_len"].cpu().tolist() == ref["enc_len"].tolist()
    top2 = ref["logp"].topk(2, -1).values
    clear = (top2[..., 0] - top2[..., 1]) > 2 * logp_tol(ref["logp"])
    assert (r["pred"].cpu()[clear] == ref["pred"][clear]).all()
    assert clear.float().mean() > 0.95


@pytest.mark.parametrize("model,B,L", [("quartznet12x1_vi", 3, 20321), ("quartznet15x5", 2, 48000), ("quartznet12x1_vi", 1, 257)])
def test_no_writes_outside_the_workspace_and_the_outputs(gpu, model, B, L):
    """Every buffer the fused call writes sits between two guard regions holding a sentinel; nothing outside the
    declared sizes (vasr_workspace_bytes, [B][T'] outputs) may change.  The scratch itself starts filled with NaNs and
    the results must equal an ordinary call's bit for bit: no kernel consumes scratch it (or a predecessor) has not
    written, padding columns included."""
    from viet_asr_amd import _lib, configs, synth
    cfg = configs.builtin(model)
    jas = cfg["JasperEncoder"]["jasper"]
    V1 = len(cfg["labels"]) + 1
    eng = _engine(cfg, synth.encoder_state_dict(jas, 64, 7), synth.decoder_state_dict(1024, V1, 7))
    sig, lens = synth.audio_batch(B, L, 7, ragged=B > 1)
    wav, ln = torch.from_numpy(sig).to(gpu), torch.from_numpy(lens).to(gpu)
    _, t1 = eng.frames(L)
    G = 1 << 16                                            # guard bytes on each side

    def guarded(nbytes):
        buf = torch.full((((nbytes + 15) // 16) * 16 + 2 * G,), 0xA5, dtype=torch.uint8, device=gpu)
        return buf, buf[G : G + nbytes]

    need = eng.handle.workspace_bytes(B, samples=L)
    bufs = {k: guarded(n) for k, n in dict(ws=need, ids=B * t1 * 4, id_len=B * 4, pred=B * t1 * 8, enc_len=B * 4,
                                           logp=B * t1 * V1 * 4).items()}
    bufs["ws"][1].fill_(0xFF)        # every float of the scratch starts as NaN: nothing may be read before it is written
    p = {k: v[1].data_ptr() for k, v in bufs.items()}
    _lib.check(_lib.lib().vasr_transcribe_greedy_f32(eng.handle.h, wav.data_ptr(), ln.data_ptr(), B, L, p["pred"], p["ids"],
                                                     p["id_len"], p["logp"], p["enc_len"], p["ws"], need,
                                                     torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    for k, (whole, inner) in bufs.items():
        n = inner.numel()
        assert bool((whole[:G] == 0xA5).all()) and bool((whole[G + n :] == 0xA5).all()), f"write outside {k}"
    ref = eng.forward(wav, ln, want_logp=True)             # and the guarded call computed the same thing
    assert torch.equal(bufs["ids"][1].view(torch.int32).view(B, t1)[0, : int(ref["id_len"][0])], ref["ids"][0, : int(ref["id_len"][0])])
    assert torch.equal(bufs["logp"][1].view(torch.float32).view(B, t1, V1), ref["logp"])


@pytest.mark.parametrize("name", REAL_AUDIO_CASES)
def test_real_recordings_through_vietasr(gpu, tmp_path, name):
    """BASELINE config 1 plumbing on the GPU: the int16 samples of a recording from the reference's audio_samples/ go
    through ``VietASR.transcribe`` (infer.py:167-171; the 8 kHz file resampled on the device as librosa.load(sr=16000)
    does on the host, infer.py:200) with checkpoints written in the reference's state_dict format, and come out as the
    imported reference produced them: the real Vietnamese head (blank on every frame of an untrained encoder's features:
    empty transcript, log-probs compared) and a seeded head (non-trivial transcript)."""
    from viet_asr_amd.infer import VietASR
    g, cfg, pcm, sr, enc_sd, real_head, syn_head = load_real_audio_golden(name)
    enc_p = str(tmp_path / "JasperEncoder-STEP-1.pt")
    torch.save({k: torch.as_tensor(v) for k, v in enc_sd.items()}, enc_p)
    for tag, head, logp_key, hyp_key in (("real", real_head, "logp", "hyp"), ("seeded", syn_head, "logp_syn", "hyp_syn")):
        dec_p = str(tmp_path / f"JasperDecoderForCTC-{tag}.pt")
        torch.save({k: torch.as_tensor(v) for k, v in head.items()}, dec_p)
        asr = VietASR("quartznet12x1_vi", enc_p, dec_p, device="gpu", decoder="greedy")
        assert asr.transcribe(pcm, sample_rate=sr) == str(g[hyp_key][0])
        # log-probs through the fused engine, fed with the signal the fixture's reference run saw: for the 8 kHz file that
        # is the ORACLE resampler's output (the device resampler agrees with it to 2e-6, checked here; through ~60 layers
        # that input difference alone moves log-probs of magnitude 120 by several 1e-3, which is not the model's error)
        x = asr._to_model_rate(pcm, sr)
        assert len(x) == int(g["samples16"])
        if sr != 16000:
            from oracle import audio_oracle as AO
            x_ref = AO.resample(pcm.astype(np.float32) / 32768.0, sr, 16000)
            assert float(np.abs(x - x_ref).max()) <= 2e-6
            x = x_ref
        r = asr._fused_engine().forward(torch.from_numpy(x)[None].to(gpu), torch.tensor([len(x)], device=gpu), want_logp=True)
        err = float(np.abs(r["logp"].cpu().numpy() - g[logp_key]).max())
        _record("real_audio", name=name, head=tag, err=err, scale=np.abs(g[logp_key]).max())
        # The 8 kHz recording has no energy above 4 kHz: its upper mel bins sit at the log guard, where the per-feature
        # normalisation (divide by a tiny std) amplifies FFT round-off -- ill-conditioned in the reference itself (DESIGN §2,
        # conditioning note); measured 7.5e-3 at |log-prob| 120 against 2.3e-4 for the 16 kHz file.  Ten times the bound there.
        assert err <= logp_tol(g[logp_key]) * (1 if sr == 16000 else 10), (tag, err)
        assert (r["pred"].cpu().numpy() == g["pred" if tag == "real" else "pred_syn"]).all()
        assert r["enc_len"].cpu().tolist() == g["enc_len"].tolist()


@pytest.mark.parametrize("K,dil,C,B,T,ragged", [(33, 1, 256, 5, 501, True), (39, 1, 256, 2, 130, False), (51, 1, 512, 4, 1300, True),
                                                (63, 1, 512, 3, 516, True), (75, 1, 512, 6, 501, False), (87, 2, 512, 3, 777, True),
                                                (75, 1, 512, 65, 501, True), (33, 1, 256, 21, 1030, True)])
def test_depthwise_on_the_matrix_pipe_matches_fp64_convolution(gpu, K, dil, C, B, T, ragged):
    """encoder_dw_mfma.hip (Toeplitz form, fp16-split operands) on one isolated layer against a float64 depthwise
    convolution of the masked input: utterance counts that do not fill a wavefront's walk (8, 4 or 2 utterances per
    wavefront; the last one is then computed twice), several 512-frame tiles, a last tile of one 256-frame group,
    ragged lengths (input mask, output zeroed past the length), dilation 2; error bounded like the packed-FMA kernel's
    (2e-6 of the largest output), which is run beside it where it exists (dilation 1); published maxima exact."""
    from viet_asr_amd import _lib
    L = _lib.dev_lib()      # include/vasr_devtools.h lives in the devtools build
    ld = int(L.vasr_padded_frames(T))
    g = torch.Generator().manual_seed(K * 1000 + T)
    x = torch.randn(B, C, ld, generator=g)
    x[1 % B] *= 300.0                                          # utterances at different levels
    x[:, :, T:] = float("nan")                                 # the padding columns must never be consumed
    x = x.to(gpu)
    w = (torch.randn(C, K, generator=g) / K ** 0.5).contiguous()
    w[3] *= 1e-3
    lens = torch.full((B,), T, dtype=torch.int32)
    if ragged:
        lens = torch.randint(1, T + 1, (B,), generator=g, dtype=torch.int32)
        lens[0] = T
    tsz = int(L.vasr_depthwise_mfma_table_size(K, dil))
    assert tsz > 0
    tab, inv = torch.empty(C, tsz, dtype=torch.int32), torch.empty(C)
    _lib.check(L.vasr_pack_depthwise_taps(w.data_ptr(), C, K, dil, tab.data_ptr(), inv.data_ptr()), L)
    y = torch.full((B, C, ld), float("nan"), device=gpu)
    stride = max(256, C * ((ld + 255) // 256) * 4)
    amax = torch.full((2, B, stride), -1, dtype=torch.int32, device=gpu)    # per-wavefront maxima tables (x, y)
    st = torch.cuda.current_stream().cuda_stream
    lens_d, tab_d, inv_d = lens.to(gpu), tab.to(gpu), inv.to(gpu)
    _lib.check(L.vasr_bench_depthwise_mfma(x.data_ptr(), tab_d.data_ptr(), inv_d.data_ptr(), lens_d.data_ptr(), B, C, T, K, dil,
                                           y.data_ptr(), amax.data_ptr(), stride, st), L)
    torch.cuda.synchronize()
    pad = (dil * K) // 2 - 1 if dil > 1 else K // 2
    t = torch.arange(ld, device=gpu)
    valid = (t[None, :] < lens_d[:, None].long())[:, None, :]
    xm = torch.where(valid, torch.nan_to_num(x, nan=0.0), torch.zeros((), device=gpu)).double()
    ref = torch.nn.functional.conv1d(xm, w.double().to(gpu)[:, None, :], padding=pad, dilation=dil, groups=C)
    t_out = ref.shape[-1]
    ref = torch.where(valid[..., :t_out], ref, torch.zeros((), device=gpu, dtype=torch.float64))
    assert bool(torch.isfinite(y).all()), "a column below the row pitch was not written (or padding was consumed)"
    for b in range(B):
        scale = float(ref[b].abs().max())
        err = float((y[b, :, :t_out].double() - ref[b]).abs().max())
        _record("dw_mfma", K=K, dil=dil, b=b, err=err, scale=scale)
        assert err <= 2e-6 * max(scale, 1e-30), (K, dil, b, err, scale)
        assert float(y[b, :, int(lens[b]):].abs().max() if int(lens[b]) < ld else 0.0) == 0.0        # zero past the length
    got = amax.view(torch.float32).amax(-1).cpu()
    assert torch.equal(got[0], xm.abs().amax((1, 2)).float().cpu())
    assert torch.equal(got[1], y.abs().amax((1, 2)).cpu())
    if dil == 1:                                              # the packed-FMA kernel on the same layer, for the record
        y2 = torch.empty(B, C, ld, device=gpu)
        x0, w_d = torch.nan_to_num(x, nan=0.0), w.to(gpu)           # named: a temporary's storage may be reused before the launch
        _lib.check(L.vasr_bench_depthwise(x0.data_ptr(), w_d.data_ptr(), lens_d.data_ptr(), B, C, T, K, y2.data_ptr(), st), L)
        torch.cuda.synchronize()
        assert float((y2[:, :, :t_out].double() - ref).abs().max()) <= 2e-6 * float(ref.abs().max())


@pytest.mark.parametrize("gemm", ["f16x2", "bf16x3", "fp32"])
def test_long_recording_in_bounded_memory_equals_the_one_pass_result(gpu, gemm):
    """engine.forward_long: a five-minute recording through windows of 2048 output frames + receptive-field halo (636 mel
    frames each side for QuartzNet12x1), three windows per pass -- every kept frame sees exactly the inputs of the
    one-pass computation, so predictions and ids are identical, and so are the log-probs BIT FOR BIT in the fp32 and
    3 x bf16 arithmetics.  In the default 2 x fp16 arithmetic the power-of-two operand scale follows the maximum of the
    row a kernel works on -- a window here, the whole recording there -- and samples more than 2^17 below that maximum
    keep fewer bits: the log-probs agree to 1e-4 (measured 8.8e-5 at |log-prob| 125), inside the tolerance against the
    reference.  The workspace is bounded by the window, not by the recording (the reference CLI refuses anything longer
    than 10 s, infer.py:201-203)."""
    from viet_asr_amd import configs, synth
    cfg = configs.builtin("quartznet12x1_vi")
    jas = cfg["JasperEncoder"]["jasper"]
    eng = _engine(cfg, synth.encoder_state_dict(jas, 64, 4), synth.decoder_state_dict(1024, 91, 4), gemm)
    n = 5 * 60 * 16000 + 123
    x = torch.from_numpy((0.1 * np.random.default_rng(5).standard_normal(n)).astype(np.float32)).to(gpu)
    x[: n // 3] *= 0.05                                              # level changes along the recording
    one = eng.forward(x[None], torch.tensor([n], device=gpu), want_logp=True)
    r = eng.forward_long(x, chunk_frames=2048, rows_per_pass=3, want_logp=True)
    assert eng.halo_mel_frames() == 636
    assert r["logp"].shape == one["logp"].shape == (1, 15001, 91)
    assert torch.equal(r["pred"], one["pred"])
    assert torch.equal(r["id_len"], one["id_len"]) and torch.equal(r["ids"][0, : int(r["id_len"][0])], one["ids"][0, : int(one["id_len"][0])])
    assert float(r["enc_len"][0]) == float(one["enc_len"][0])
    _record("long_chunked", gemm=gemm, err=(r["logp"] - one["logp"]).abs().max(), scale=one["logp"].abs().max())
    if gemm == "f16x2":
        assert float((r["logp"] - one["logp"]).abs().max()) <= logp_tol(one["logp"].cpu().numpy())
    else:
        assert torch.equal(r["logp"], one["logp"])
    # bounded: three windows of 2 * 2048 + 2 * 636 mel frames instead of 30 001 frames
    assert r["workspace_bytes"] < 0.6 * r["one_pass_workspace_bytes"]
    hour = eng.handle.workspace_bytes(1, samples=3600 * 16000)
    assert r["workspace_bytes"] < hour / 20
    # a recording shorter than one window degenerates to a single full-length row
    short = eng.forward_long(x[:48000], chunk_frames=2048, rows_per_pass=3)
    one_s = eng.forward(x[None, :48000].contiguous(), torch.tensor([48000], device=gpu))
    assert torch.equal(short["pred"], one_s["pred"])
