"""Host-side constants of the mel front end: window and mel filterbank.

In the reference these are module buffers built in FilterbankFeatures.__init__
(nemo/collections/asr/parts/features.py:171-205): ``torch.hann_window(win, periodic=False)`` and
``librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax)``.  librosa is a third-party package that is
neither in the reference tree nor in this image, so the Slaney-scale filterbank is rebuilt here
from its published definition (float64 maths, float32 result), as librosa does with
``htk=False, norm='slaney'``.
"""
import math

import numpy as np
import torch


def mel_filterbank(sr=16000, n_fft=512, n_mels=64, fmin=0.0, fmax=None, variant="librosa"):
    """Triangular mel filters on the Slaney scale with area normalisation; [n_mels, n_fft//2+1] f32.

    variant="librosa" (default) reproduces the order of roundings of ``librosa.filters.mel`` (< 0.10, the API the reference's
    positional call needs): the un-normalised triangle is stored into a FLOAT32 array, then multiplied in place by the float64
    normalisers ``2 / (f[i+2] - f[i])`` -- two roundings.  variant="f64" builds both in float64 and rounds once (what rounds
    1-5 shipped): 1 ulp apart in 140 of the 498 non-zero coefficients of the (16000, 512, 64, 0, 8000) bank."""
    if variant not in ("librosa", "f64"):
        raise ValueError("variant: 'librosa' or 'f64'")
    fmax = float(sr) / 2 if fmax is None else float(fmax)
    lin_step = 200.0 / 3.0            # Hz per mel below 1 kHz
    brk_hz = 1000.0
    brk_mel = brk_hz / lin_step
    log_step = math.log(6.4) / 27.0   # mel step above 1 kHz

    def to_mel(hz):
        return hz / lin_step if hz < brk_hz else brk_mel + math.log(hz / brk_hz) / log_step

    def to_hz(mel):
        return lin_step * mel if mel < brk_mel else brk_hz * math.exp(log_step * (mel - brk_mel))

    edges = np.array([to_hz(m) for m in np.linspace(to_mel(float(fmin)), to_mel(fmax), n_mels + 2)])
    bins = np.linspace(0.0, float(sr) / 2, n_fft // 2 + 1)
    fb = np.zeros((n_mels, bins.size), dtype=np.float32 if variant == "librosa" else np.float64)
    for i in range(n_mels):
        lo, ce, hi = edges[i], edges[i + 1], edges[i + 2]
        rise = (bins - lo) / (ce - lo)
        fall = (hi - bins) / (hi - ce)
        fb[i] = np.maximum(0.0, np.minimum(rise, fall))          # "librosa": rounded to float32 here ...
    fb *= (2.0 / (edges[2:] - edges[:-2]))[:, None]               # ... and again after the float64 product
    return fb.astype(np.float32)


def frontend_description(pre_cfg):
    """AudioToMelSpectrogramPreprocessor kwargs (audio_preprocessing.py:314-373) -> library front-end dict.

    Mirrors the constructor's argument handling and its ValueErrors; dither / pad_to are the module's business
    (asr.AudioToMelSpectrogramPreprocessor).
    """
    c = dict(pre_cfg)
    sr = int(c.get("sample_rate", 16000))
    ws, wst = c.get("window_size", 0.02), c.get("window_stride", 0.01)
    nws, nwst = c.get("n_window_size"), c.get("n_window_stride")
    if ws and nws:
        raise ValueError("received both window_size and n_window_size. Only one should be specified.")
    if wst and nwst:
        raise ValueError("received both window_stride and n_window_stride. Only one should be specified.")
    if ws:
        nws = int(ws * sr)
    if wst:
        nwst = int(wst * sr)
    if not isinstance(nws, int) or not isinstance(nwst, int) or nws <= 0 or nwst <= 0:
        raise ValueError("got an invalid value for either n_window_size or n_window_stride. "
                         "Both must be positive ints.")
    n_fft = c.get("n_fft") or 2 ** math.ceil(math.log2(nws))
    # stft_conv=True (configs/quartznet15x5.yaml:26) routes the STFT through the third-party torch_stft package
    # (features.py:155-166), neither vendored nor installed: "parity unpinned".  What that package publishes is the same
    # centred, reflect-padded DFT computed as a conv1d with a Fourier basis, windowed by
    # scipy.signal.get_window(window, win_length, fftbins=True) -- the PERIODIC window, where torch.stft gets the
    # symmetric one (features.py:178) -- and returned as a magnitude that features.py:260-261 squares again.  The kernels
    # therefore run the same transform with the periodic window (power = re^2 + im^2 instead of sqrt(.)^2: <= 1 ulp).
    periodic = bool(c.get("stft_conv", False))
    guard_type = c.get("log_zero_guard_type", "add")
    if guard_type not in ("add", "clamp"):
        raise ValueError(f"{type(c).__name__} received {guard_type} for the log_zero_guard_type parameter. "
                         "It must be either 'add' or 'clamp'.")                       # features.py:215-220
    if c.get("log", True) is not True:
        raise NotImplementedError("only log=True is implemented")
    if float(c.get("mag_power", 2.0)) != 2.0 or int(c.get("frame_splicing", 1)) != 1:
        raise NotImplementedError("only mag_power=2, frame_splicing=1 are implemented")
    guard = c.get("log_zero_guard_value", 2 ** -24)
    if isinstance(guard, str):
        guard = {"tiny": torch.finfo(torch.float32).tiny, "eps": torch.finfo(torch.float32).eps}[guard]
    window = c.get("window", "hann")
    fns = {"hann": torch.hann_window, "hamming": torch.hamming_window, "blackman": torch.blackman_window,
           "bartlett": torch.bartlett_window}
    if window not in fns:
        raise NotImplementedError(f"window {window!r} is not implemented")
    win = fns[window](nws, periodic=periodic).to(torch.float32).numpy()
    n_mels = int(c.get("features", 64))
    fb = mel_filterbank(sr, n_fft, n_mels, c.get("lowfreq", 0) or 0.0, c.get("highfreq") or sr / 2)
    norm = c.get("normalize", "per_feature")
    if norm not in ("per_feature", "all_features", None, False, ""):
        raise NotImplementedError(f"normalize={norm!r} is not implemented (per_feature, all_features or none)")
    if float(c.get("pad_value", 0)) != 0.0:
        raise NotImplementedError("pad_value other than 0 is not implemented")
    return dict(sample_rate=sr, n_fft=int(n_fft), win_length=nws, hop_length=nwst, n_mels=n_mels,
                preemph=c.get("preemph", 0.97), log_guard=float(guard),
                log_guard_type=guard_type, normalize=norm if norm in ("per_feature", "all_features") else "none", window=win,
                filterbank=fb)
