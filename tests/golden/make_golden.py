#!/usr/bin/env python3
"""Generate tests/golden/*.npz by running the REAL reference modules (dev container only).

    python tests/golden/make_golden.py            # needs /root/reference

The reference (vendored NeMo 0.10 under /root/reference/nemo) is imported unchanged; the
pip packages it wants but this image lacks are replaced by import-time stubs (SURVEY.md
§8c).  The only stub that carries arithmetic is ``librosa.filters.mel`` -> our Slaney
restatement (oracle.quartznet_oracle.slaney_mel_filterbank): that filterbank is third-party
code absent from the reference tree, hence "parity unpinned" for A4.

What is recorded per case (small, KB-sized): inputs are NOT stored -- they are regenerated
from viet-asr_amd/synth.py seeds; outputs: mel, seq, enc_len, log_probs, predictions,
transcripts, a strided slice + float64 checksum of the encoder output.
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch
import yaml

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = "/root/reference"
sys.path.insert(0, REPO)


def _load_pkg():
    import viet_asr_amd  # noqa: F401  (root shim -> viet-asr_amd/)
    return sys.modules["viet_asr_amd"]


def install_shims():
    for a, v in (("int", int), ("float", float), ("str", str), ("bool", bool), ("object", object)):
        if a not in np.__dict__:
            setattr(np, a, v)
    if not hasattr(np, "sctypes"):
        np.sctypes = {"int": [np.int8, np.int16, np.int32, np.int64],
                      "uint": [np.uint8, np.uint16, np.uint32, np.uint64],
                      "float": [np.float16, np.float32, np.float64],
                      "complex": [np.complex64, np.complex128], "others": [bool, object, bytes, str]}

    def stub(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    # wrapt: decorator() -> wrapper(wrapped, instance, args, kwargs)
    def _decorator(wrapper=None, enabled=None):
        def deco(wrapped):
            if isinstance(wrapped, type):
                return wrapped
            import functools

            @functools.wraps(wrapped)
            def inner(*a, **k):
                return wrapper(wrapped, None, a, k)
            return inner
        return deco

    class _FW:
        def __init__(self, wrapped, wrapper):
            self.__wrapped__, self._w = wrapped, wrapper

        def __call__(self, *a, **k):
            return self._w(self.__wrapped__, None, a, k)

        def __get__(self, inst, owner):
            import functools
            if inst is None:
                return self
            return functools.partial(self.__call__, inst)
    stub("wrapt", decorator=_decorator, FunctionWrapper=_FW)

    class _YAML:
        def __init__(self, typ=None):
            pass

        def load(self, f):
            return yaml.safe_load(f)

        def dump(self, d, f):
            return yaml.safe_dump(d, f)
    ry = stub("ruamel")
    ry.yaml = stub("ruamel.yaml", YAML=_YAML)

    from oracle.quartznet_oracle import slaney_mel_filterbank
    lib = stub("librosa")
    lib.filters = stub("librosa.filters", mel=lambda sr, n_fft, n_mels=128, fmin=0.0, fmax=None:
                       slaney_mel_filterbank(sr, n_fft, n_mels, fmin, fmax))
    lib.core = stub("librosa.core")
    lib.effects = stub("librosa.effects")

    class _Dummy:
        def __init__(self, *a, **k):
            pass
    stub("torch_stft", STFT=_Dummy)
    stub("pyctcdecode", build_ctcdecoder=lambda *a, **k: None)
    for n in ("kenlm", "soundfile", "kaldi_io", "inflect", "unidecode", "frozendict", "wget", "sox",
              "wandb", "loguru", "torchvision", "torchvision.transforms", "torchvision.datasets",
              "torchaudio", "apex", "braceexpand", "webdataset", "editdistance", "numba"):
        if n in sys.modules and sys.modules[n].__spec__ is None:
            continue
        try:
            have = importlib.util.find_spec(n.split(".")[0]) is not None
        except ValueError:
            have = False
        if not have or n.startswith("torchvision"):
            stub(n)
    sys.modules["frozendict"].frozendict = dict
    sys.modules["unidecode"].unidecode = lambda s: s
    sys.modules["inflect"].engine = _Dummy
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    sys.modules["torchvision"].datasets = sys.modules["torchvision.datasets"]
    del sys.modules["torchaudio"]  # reference guards it with try/except ModuleNotFoundError

    # legacy real-valued torch.stft (parts/features.py:181-188 passes no return_complex)
    _stft = torch.stft

    def stft_legacy(x, *a, **k):
        if "return_complex" in k:
            return _stft(x, *a, **k)
        return torch.view_as_real(_stft(x, *a, return_complex=True, **k))
    torch.stft = stft_legacy


def build_reference(cfg, labels, pre_overrides=None):
    import nemo
    import nemo.collections.asr as nemo_asr
    nf = nemo.core.NeuralModuleFactory(placement=nemo.core.DeviceType.CPU)
    pre_cfg = dict(cfg.get("AudioToMelSpectrogramPreprocessor") or cfg["AudioPreprocessing"])
    pre_cfg["dither"] = 0          # infer.py:89
    pre_cfg["pad_to"] = 0          # infer.py:90
    pre_cfg["stft_conv"] = False   # 15x5 yaml says true (torch_stft, absent); run like the vi config
    pre_cfg.pop("feat_type", None)
    pre_cfg.update(pre_overrides or {})   # constructor options no shipped YAML uses (round 6: log guard "clamp", all_features)
    pre = nemo_asr.AudioToMelSpectrogramPreprocessor(**pre_cfg)
    enc = nemo_asr.JasperEncoder(feat_in=pre_cfg["features"], **cfg["JasperEncoder"])
    dec = nemo_asr.JasperDecoderForCTC(feat_in=cfg["JasperEncoder"]["jasper"][-1]["filters"],
                                       num_classes=len(labels))
    greedy = nemo_asr.GreedyCTCDecoder()
    return nf, pre, enc, dec, greedy


def run_case(name, cfg_file, batch, samples, seed, ragged, real_decoder=None, band_hz=0, pre_overrides=None):
    pkg = _load_pkg()
    synth = pkg.synth
    from nemo.collections.asr.helpers import post_process_predictions
    cfg = yaml.safe_load(open(os.path.join(REF, "configs", cfg_file), encoding="utf-8"))
    labels = cfg["labels"]
    nf, pre, enc, dec, greedy = build_reference(cfg, labels, pre_overrides)
    jas = cfg["JasperEncoder"]["jasper"]
    enc_sd = synth.encoder_state_dict(jas, 64, seed)
    if real_decoder:
        dec_sd = {k: v.numpy() for k, v in torch.load(os.path.join(REF, real_decoder), map_location="cpu").items()}
    else:
        dec_sd = synth.decoder_state_dict(jas[-1]["filters"], len(labels) + 1, seed)
    missing = set(enc.state_dict().keys()) ^ set(enc_sd.keys())
    assert not missing, sorted(missing)[:8]
    enc.load_state_dict({k: torch.as_tensor(v) for k, v in enc_sd.items()})
    dec.load_state_dict({k: torch.as_tensor(v) for k, v in dec_sd.items()})
    sig, lens = synth.audio_batch(batch, samples, seed, ragged, band_hz=band_hz or None)
    # the executor's call convention: pmodule(force_pt=True, **ports)  (actions.py:419-428)
    enc.eval(); dec.eval(); greedy.eval()                      # actions.py:412-415 (nn.Modules only: Q1)
    with torch.no_grad():
        if pre_overrides and pre_overrides.get("dither"):
            torch.manual_seed(seed)     # features.py:250-251 draws torch.randn_like(x) from the global CPU generator
            sig = sig.copy()            # ... and adds it IN PLACE to the tensor that shares the array's memory
        mel, seq = pre(force_pt=True, input_signal=torch.as_tensor(sig), length=torch.as_tensor(lens))
        e, elen = enc(force_pt=True, audio_signal=mel, length=seq)
        logp = dec(force_pt=True, encoder_output=e)
        pred = greedy(force_pt=True, log_probs=logp)
    hyp = post_process_predictions([pred], labels)
    top2 = torch.topk(logp, 2, dim=-1).values
    margin = (top2[..., 0] - top2[..., 1])
    out = dict(
        cfg_file=cfg_file, batch=batch, samples=samples, seed=seed, ragged=ragged,
        real_decoder=real_decoder or "",
        lens=lens, mel=mel.numpy(), seq=seq.numpy(), enc_len=elen.numpy(),
        enc_slice=e[:, ::37, ::5].numpy().copy(), enc_sum=np.float64(e.double().sum().item()),
        enc_abs_sum=np.float64(e.double().abs().sum().item()),
        logp=logp.numpy(), pred=pred.numpy(), hyp=np.array(hyp, dtype=object).astype("U"),
        min_margin=np.float32(margin.min().item()),
        fb=pre.filter_banks[0].numpy(),
    )
    if pre_overrides:
        import json
        out["pre_overrides"] = json.dumps(pre_overrides, sort_keys=True)
    if band_hz:   # (only the round-5 cases carry these keys: the older fixtures regenerate bit-identically without them)
        out["band_hz"] = band_hz
        out["sig_abs_sum"] = np.float64(np.abs(sig.astype(np.float64)).sum())
        # the band-limited input itself (np.sinc / np.kaiser / the convolution's summation order are not bit-
        # reproducible across numpy / libm builds): a host that regenerates it differently replays the stored one
        out["sig"], out["sig_lens"] = sig.astype(np.float32), np.asarray(lens, dtype=np.int64)
        out["margin"] = margin.numpy()          # per frame: the tests may excuse a frame only by ITS margin
        if real_decoder:
            # the shipped head answers blank on every frame of an untrained encoder (empty transcripts): the same encoder output
            # also goes through a seeded head whose transcripts bite (as run_real_audio_case does)
            with torch.no_grad():
                dec.load_state_dict({k: torch.as_tensor(v) for k, v in synth.decoder_state_dict(jas[-1]["filters"], len(labels) + 1, seed).items()})
                logp_syn = dec(force_pt=True, encoder_output=e)
                pred_syn = greedy(force_pt=True, log_probs=logp_syn)
            t2 = torch.topk(logp_syn, 2, dim=-1).values
            out["logp_syn"] = logp_syn.numpy()
            out["pred_syn"] = pred_syn.numpy()
            out["margin_syn"] = (t2[..., 0] - t2[..., 1]).numpy()
            out["hyp_syn"] = np.array(post_process_predictions([pred_syn], labels), dtype=object).astype("U")
            print("   hyp_syn[0][:60] =", repr(str(out["hyp_syn"][0])[:60]), f"min_margin_syn={out['margin_syn'].min():.3e}")
    if real_decoder:  # the shipped CTC-head checkpoint is data; carry it so the GPU box can replay the case
        out["dec_weight"] = dec_sd["decoder_layers.0.weight"]
        out["dec_bias"] = dec_sd["decoder_layers.0.bias"]
    path = os.path.join(REPO, "tests", "golden", name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: mel{tuple(mel.shape)} enc{tuple(e.shape)} enc_len={elen.tolist()} "
          f"min_margin={margin.min().item():.3e} bytes={os.path.getsize(path)}")
    print("   hyp[0][:60] =", repr(hyp[0][:60]))


def run_real_audio_case(name, wav_name, seed):
    """One file of the reference's audio_samples/ through the imported reference, the way infer.py:194-206 feeds it:
    decode to float32 in [-1, 1) at 16 kHz (librosa.load(sr=16000), infer.py:200 -- third-party: int16 / 2^15 is what it
    returns for PCM, and its 8 kHz -> 16 kHz conversion is resampy's kaiser_best, restated in oracle/audio_oracle.py,
    parity unpinned), one utterance per call, synthetic encoder weights (the trained encoder is not in the mount) and
    the REAL Vietnamese CTC head.  The fixture stores the file's int16 samples (data) and the reference's outputs."""
    import wave
    pkg = _load_pkg()
    synth = pkg.synth
    from nemo.collections.asr.helpers import post_process_predictions
    from oracle import audio_oracle as AO
    with wave.open(os.path.join(REF, "audio_samples", wav_name)) as w:
        assert w.getnchannels() == 1 and w.getsampwidth() == 2
        sr = w.getframerate()
        pcm = np.frombuffer(w.readframes(w.getnframes()), dtype="<i2").copy()
    x = pcm.astype(np.float32) / 32768.0
    if sr != 16000:
        x = AO.resample(x, sr, 16000)
    cfg = yaml.safe_load(open(os.path.join(REF, "configs", "quartznet12x1_vi.yaml"), encoding="utf-8"))
    labels = cfg["labels"]
    nf, pre, enc, dec, greedy = build_reference(cfg, labels)
    jas = cfg["JasperEncoder"]["jasper"]
    enc_sd = synth.encoder_state_dict(jas, 64, seed)
    dec_sd = {k: v.numpy() for k, v in torch.load(os.path.join(REF, "models/acoustic_model/vietnamese/JasperDecoderForCTC-STEP-289936.pt"),
                                                  map_location="cpu").items()}
    enc.load_state_dict({k: torch.as_tensor(v) for k, v in enc_sd.items()})
    dec.load_state_dict({k: torch.as_tensor(v) for k, v in dec_sd.items()})
    enc.eval(); dec.eval(); greedy.eval()
    with torch.no_grad():
        mel, seq = pre(force_pt=True, input_signal=torch.as_tensor(x)[None], length=torch.tensor([len(x)]))
        e, elen = enc(force_pt=True, audio_signal=mel, length=seq)
        logp = dec(force_pt=True, encoder_output=e)
        pred = greedy(force_pt=True, log_probs=logp)
        # On untrained encoder features the real head answers blank on every frame (empty transcript): the same encoder
        # output also goes through a seeded head, whose non-trivial transcript makes the comparison bite.
        dec.load_state_dict({k: torch.as_tensor(v) for k, v in synth.decoder_state_dict(1024, len(labels) + 1, seed).items()})
        logp_syn = dec(force_pt=True, encoder_output=e)
        pred_syn = greedy(force_pt=True, log_probs=logp_syn)
    hyp = post_process_predictions([pred], labels)
    hyp_syn = post_process_predictions([pred_syn], labels)
    margin = lambda lp: np.float32((torch.topk(lp, 2, dim=-1).values[..., 0] - torch.topk(lp, 2, dim=-1).values[..., 1]).min().item())
    # the real head's weights are in vi12x1_b2_q2_realdec.npz (same checkpoint): not stored twice
    out = dict(cfg_file="quartznet12x1_vi.yaml", wav_name=wav_name, sample_rate=sr, pcm=pcm, seed=seed, samples16=len(x),
               seq=seq.numpy(), enc_len=elen.numpy(), logp=logp.numpy(), pred=pred.numpy(),
               hyp=np.array(hyp, dtype=object).astype("U"), min_margin=margin(logp),
               logp_syn=logp_syn.numpy(), pred_syn=pred_syn.numpy(), hyp_syn=np.array(hyp_syn, dtype=object).astype("U"),
               min_margin_syn=margin(logp_syn),
               mel_sum=np.float64(mel.double().sum().item()), mel_slice=mel[:, ::7, ::11].numpy().copy())
    path = os.path.join(REPO, "tests", "golden", name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: {wav_name} sr={sr} pcm={len(pcm)} -> mel{tuple(mel.shape)} enc_len={elen.tolist()} "
          f"min_margin={out['min_margin']:.3e} bytes={os.path.getsize(path)}")
    print("   hyp =", repr(hyp[0][:80]), " hyp_syn =", repr(hyp_syn[0][:60]), f" min_margin_syn={out['min_margin_syn']:.3e}")


if __name__ == "__main__":
    assert os.path.isdir(REF), "reference checkout required (dev container only)"
    install_shims()
    sys.path.insert(0, REF)
    torch.manual_seed(0)
    VI_DEC = "models/acoustic_model/vietnamese/JasperDecoderForCTC-STEP-289936.pt"
    run_case("vi12x1_b3_ragged", "quartznet12x1_vi.yaml", 3, 32480, 1, True)
    run_case("vi12x1_b2_q2_realdec", "quartznet12x1_vi.yaml", 2, 24000, 2, True, real_decoder=VI_DEC)
    run_case("en15x5_b2_ragged", "quartznet15x5.yaml", 2, 20321, 3, True)
    run_case("vi12x1_b1_tiny", "quartznet12x1_vi.yaml", 1, 4000, 4, False)
    run_case("en12x1_b4_hopmult", "quartznet12x1.yaml", 4, 160 * 150, 5, True)     # third shipped config; L % hop == 0
    run_case("en15x5_b1_10s", "quartznet15x5.yaml", 1, 160000, 6, False)           # one full BASELINE-length clip
    # round 5: the band-limited regime of 8 kHz-sourced audio, batched and ragged
    run_case("vi12x1_b4_band4k_ragged", "quartznet12x1_vi.yaml", 4, 40000, 9, True, real_decoder=VI_DEC, band_hz=4000)
    run_case("en15x5_b2_band4k", "quartznet15x5.yaml", 2, 30000, 10, False, band_hz=4000)
    # round 6: front-end constructor options the reference accepts and no shipped YAML uses (parts/features.py:31-39, 272-273)
    run_case("vi12x1_b3_clamp_allfeat", "quartznet12x1_vi.yaml", 3, 24000, 12, True,
             pre_overrides={"log_zero_guard_type": "clamp", "normalize": "all_features"})
    run_case("vi12x1_b2_clamp_1e5", "quartznet12x1_vi.yaml", 2, 16000, 13, True,
             pre_overrides={"log_zero_guard_type": "clamp", "log_zero_guard_value": 1e-5})
    # dither (parts/features.py:250-251): noise of 1e-3 from torch's CPU generator seeded with the case's seed
    run_case("vi12x1_b2_dither_1e3", "quartznet12x1_vi.yaml", 2, 16000, 14, True, pre_overrides={"dither": 1e-3})
    # BASELINE config 1 plumbing: real recordings of the reference's audio_samples/ (16 kHz broadcast, 8 kHz call centre)
    run_real_audio_case("real16k_thoisu_5", "V1 1 11 12H00 THOI SU 2019_5.wav", 7)
    run_real_audio_case("real8k_external_2", "external_1202_771_20191118_093137_1574044304_22681_2.wav", 8)
