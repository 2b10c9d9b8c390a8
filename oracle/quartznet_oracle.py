"""CPU oracle for the viet-asr infer.py hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this file; the shipped path (viet-asr_amd/) never does and fails loudly when
its HIP library is missing.

It restates, op for op, what the reference computes on the path
AudioToMelSpectrogramPreprocessor -> JasperEncoder -> JasperDecoderForCTC ->
GreedyCTCDecoder -> CTC collapse, using the same ATen CPU primitives the
reference itself calls (torch.stft / conv1d / batch_norm / log_softmax), so its
arithmetic is the reference's arithmetic.  Every function cites the reference
file:line it follows (paths under /root/reference).

PINNING.  tests/golden/*.npz were produced by tests/golden/make_golden.py, which
imports the real reference modules in the dev container and records their
outputs; tests/test_oracle_golden.py checks this file against them.  Two parts
of the path are arithmetic in third-party packages that are NOT in the
reference tree and NOT installed here, so they are *parity unpinned*:
  * the mel filterbank (librosa.filters.mel, unpinned, call site
    parts/features.py:199-205) -- restated below from librosa's published
    Slaney-scale algorithm;
  * beam search (pyctcdecode + kenlm, requirements.txt:16) -- see beam_oracle.py;
  * the ``stft_conv=True`` STFT (torch_stft, requirements.txt, call site
    parts/features.py:155-166; configs/quartznet15x5.yaml:26 selects it) --
    ``torch_stft_magnitude`` below restates its published transform.
"""

import numpy as np
import torch
import torch.nn.functional as F

CONSTANT = 1e-5  # parts/features.py:14


# --------------------------------------------------------------------------- A4
def slaney_mel_filterbank(sr=16000, n_fft=512, n_mels=64, fmin=0.0, fmax=None, variant="librosa"):
    """librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax) with htk=False, norm='slaney', dtype=float32.

    Call site: parts/features.py:199-205 (positional ``(sr, n_fft, ...)`` => librosa < 0.10).  Published algorithm (librosa
    0.7/0.8 ``filters.mel``): Slaney mel scale (linear below 1 kHz at 200/3 Hz per mel, log above with step log(6.4)/27),
    n_mels+2 band edges, FFT bin centres ``linspace(0, sr/2, 1+n_fft//2)``, triangular weights
    ``max(0, min(lower, upper))`` from the ramps, then area normalisation ``2 / (f[i+2] - f[i])``.

    variant="librosa" (default) follows the published ORDER OF ROUNDINGS: ``weights`` is allocated as float32, every
    un-normalised triangle row is STORED into it (first rounding), then ``weights *= enorm[:, np.newaxis]`` multiplies the
    float32 values by the float64 normalisers in float64 and rounds back into the float32 array (second rounding).
    variant="f64" is rounds 1-5's restatement -- triangle and normalisation in float64, one rounding at the end: 140 of the
    498 non-zero coefficients of the reference's (16000, 512, 64, 0, 8000) bank differ from "librosa" by 1 ulp.  Parity is
    unpinned either way (librosa absent); tools/pin_third_party.py reports which variant a real librosa matches.
    """
    if variant not in ("librosa", "f64"):
        raise ValueError("variant: 'librosa' or 'f64'")
    if fmax is None:
        fmax = sr / 2.0
    f_sp = 200.0 / 3
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0

    def hz_to_mel(f):
        f = np.asarray(f, dtype=np.float64)
        m = f / f_sp
        big = f >= min_log_hz
        return np.where(big, min_log_mel + np.log(np.maximum(f, 1e-300) / min_log_hz) / logstep, m)

    def mel_to_hz(m):
        m = np.asarray(m, dtype=np.float64)
        f = f_sp * m
        big = m >= min_log_mel
        return np.where(big, min_log_hz * np.exp(logstep * (m - min_log_mel)), f)

    fftfreqs = np.linspace(0, float(sr) / 2, int(1 + n_fft // 2), endpoint=True)
    mel_f = mel_to_hz(np.linspace(hz_to_mel(fmin), hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = np.subtract.outer(mel_f, fftfreqs)
    weights = np.zeros((n_mels, int(1 + n_fft // 2)), dtype=np.float32 if variant == "librosa" else np.float64)
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        weights[i] = np.maximum(0, np.minimum(lower, upper))      # "librosa": float64 -> float32 here
    enorm = 2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels])
    weights *= enorm[:, np.newaxis]                               # "librosa": float32 * float64 in float64 -> float32
    return weights.astype(np.float32)


# --------------------------------------------------------------------------- A1-A3
def featurizer_seq_len(length, hop):
    """parts/features.py:238-239 -- ceil(len / hop) as int64 (computed in float32)."""
    return torch.ceil(length.float() / hop).to(dtype=torch.long)


def normalize_batch_per_feature(x, seq_len):
    """parts/features.py:17-30 -- per (b, f): mean / unbiased std over [:seq_len[b]]."""
    x_mean = torch.zeros((seq_len.shape[0], x.shape[1]), dtype=x.dtype)
    x_std = torch.zeros((seq_len.shape[0], x.shape[1]), dtype=x.dtype)
    for i in range(x.shape[0]):
        x_mean[i, :] = x[i, :, : seq_len[i]].mean(dim=1)
        x_std[i, :] = x[i, :, : seq_len[i]].std(dim=1)
    x_std += CONSTANT
    return (x - x_mean.unsqueeze(2)) / x_std.unsqueeze(2)


def normalize_batch_all_features(x, seq_len):
    """parts/features.py:31-39 -- per utterance: mean / unbiased std over every bin and every frame < seq_len[b]."""
    x_mean = torch.zeros(seq_len.shape, dtype=x.dtype)
    x_std = torch.zeros(seq_len.shape, dtype=x.dtype)
    for i in range(x.shape[0]):
        x_mean[i] = x[i, :, : seq_len[i].item()].mean()
        x_std[i] = x[i, :, : seq_len[i].item()].std()
    x_std += CONSTANT
    return (x - x_mean.view(-1, 1, 1)) / x_std.view(-1, 1, 1)


def torch_stft_magnitude(x, n_fft, hop, win_length, window="hann"):
    """``torch_stft.STFT(n_fft, hop, win_length, window).transform(x)[0]`` -- PARITY UNPINNED: the package is
    third-party (pseeth/torch-stft, requirements.txt, unpinned), absent from /root/reference and from this image;
    the reference holds no test of it.  Call site parts/features.py:155-166 (STFTPatch.forward).  Published
    algorithm: forward basis = rows [Re; Im] of fft(eye(n_fft))[:n_fft//2+1], each multiplied by
    ``scipy.signal.get_window(window, win_length, fftbins=True)`` (the PERIODIC window) zero-padded symmetrically to
    n_fft; the signal is reflect-padded by n_fft//2 on both sides and convolved with the basis at stride ``hop``;
    magnitude = sqrt(re^2 + im^2).  x [B, L] f32 -> [B, n_fft//2+1, 1 + L//hop] f32."""
    from scipy.signal import get_window
    x = torch.as_tensor(x, dtype=torch.float32)
    cutoff = n_fft // 2 + 1
    basis = np.fft.fft(np.eye(n_fft))
    basis = np.vstack([np.real(basis[:cutoff]), np.imag(basis[:cutoff])])
    win = get_window(window, win_length, fftbins=True)
    lpad = (n_fft - win_length) // 2                                   # librosa.util.pad_center
    win = np.pad(win, (lpad, n_fft - win_length - lpad))
    fwd = torch.FloatTensor(basis[:, None, :]) * torch.from_numpy(win).float()
    xp = F.pad(x.unsqueeze(1).unsqueeze(1), (n_fft // 2, n_fft // 2, 0, 0), mode="reflect").squeeze(1)
    t = F.conv1d(xp, fwd, stride=hop, padding=0)
    return torch.sqrt(t[:, :cutoff] ** 2 + t[:, cutoff:] ** 2)


def melspec_forward(x, length, sample_rate=16000, n_window_size=320, n_window_stride=160, n_fft=512,
                    preemph=0.97, nfilt=64, lowfreq=0, highfreq=None, log_zero_guard_value=2 ** -24,
                    normalize="per_feature", pad_value=0.0, fb=None, stft_conv=False, log_zero_guard_type="add", dither=0.0):
    """FilterbankFeatures.forward (parts/features.py:245-301) with dither=0 (default), pad_to=0
    (infer.py:89-90; quirk Q1: the featurizer is never put in eval mode, so no pad-to-16),
    mag_power=2, log guard "add", frame_splicing=1.  stft_conv=True (unpinned, see
    torch_stft_magnitude) squares the package's magnitude and skips the re/im sum (:260-263).

    x [B, L] float32, length [B] int64 -> (mel [B, nfilt, 1 + L//hop] f32, seq_len [B] i64)
    """
    x = torch.as_tensor(x, dtype=torch.float32)
    length = torch.as_tensor(length, dtype=torch.int64)
    seq_len = featurizer_seq_len(length, n_window_stride)                      # :246
    if dither > 0:                                                             # :250-251 (the caller seeds torch; not in place here)
        x = x + dither * torch.randn_like(x)
    if preemph is not None:                                                    # :254-255
        x = torch.cat((x[:, 0].unsqueeze(1), x[:, 1:] - preemph * x[:, :-1]), dim=1)
    if stft_conv:
        p = torch_stft_magnitude(x, n_fft, n_window_stride, n_window_size).pow(2.0)     # :155-166, :260-261
    else:
        window = torch.hann_window(n_window_size, periodic=False)              # :179-180
        # :181-188 torch.stft(center=True) -> reflect pad n_fft//2, window centred in n_fft
        spec = torch.stft(x, n_fft=n_fft, hop_length=n_window_stride, win_length=n_window_size,
                          center=True, window=window, return_complex=True, pad_mode="reflect")
        spec = torch.view_as_real(spec)                                        # legacy [B,F,T,2]
        p = spec.pow(2.0).sum(-1)                                              # :260-263
    if fb is None:
        fb = slaney_mel_filterbank(sample_rate, n_fft, nfilt, lowfreq, highfreq or sample_rate / 2)
    fb = torch.as_tensor(fb, dtype=torch.float32).unsqueeze(0)
    m = torch.matmul(fb, p)                                                    # :266
    if log_zero_guard_type == "add":                                           # :269-271
        m = torch.log(m + log_zero_guard_value)
    elif log_zero_guard_type == "clamp":                                       # :272-273
        m = torch.log(torch.clamp(m, min=log_zero_guard_value))
    else:
        raise ValueError("log_zero_guard_type was not understood")             # :274-275
    if normalize == "per_feature":                                             # :282-283
        m = normalize_batch_per_feature(m, seq_len)
    elif normalize == "all_features":
        m = normalize_batch_all_features(m, seq_len)
    max_len = m.size(-1)                                                       # :287-291
    mask = torch.arange(max_len).expand(m.size(0), max_len) >= seq_len.unsqueeze(1)
    m = m.masked_fill(mask.unsqueeze(1), pad_value)
    return m, seq_len


# --------------------------------------------------------------------------- A5-A8
def get_same_padding(kernel_size, stride, dilation):
    """parts/jasper.py:60-65."""
    if stride > 1 and dilation > 1:
        raise ValueError("Only stride OR dilation may be greater than 1")
    if dilation > 1:
        return (dilation * kernel_size) // 2 - 1
    return kernel_size // 2


def masked_conv1d(x, lens, weight, stride=1, padding=0, dilation=1, groups=1):
    """MaskedConv1d.forward / get_seq_len (parts/jasper.py:108-132).

    lens may be float (quirk Q3): truncated with .to(long) for the mask, and the
    returned length is a *float* tensor from true division.
    """
    lens = lens.to(dtype=torch.long)
    max_len = x.size(2)
    mask = torch.arange(max_len).expand(len(lens), max_len) >= lens.unsqueeze(1)
    x = x.masked_fill(mask.unsqueeze(1), 0)
    k = weight.shape[2]
    lens = (lens + 2 * padding - dilation * (k - 1) - 1) / stride + 1
    return F.conv1d(x, weight, None, stride, padding, dilation, groups), lens


def _t(a, dtype=None):
    t = torch.as_tensor(np.asarray(a))
    return t if dtype is None else t.to(dtype)


def _bn_eval(x, sd, prefix):
    """nn.BatchNorm1d(C, eps=1e-3) in eval mode (parts/jasper.py:392)."""
    dt = x.dtype
    return F.batch_norm(x, _t(sd[prefix + ".running_mean"], dt), _t(sd[prefix + ".running_var"], dt),
                        _t(sd[prefix + ".weight"], dt), _t(sd[prefix + ".bias"], dt), False, 0.1, 1e-3)


def _first(v):
    return v[0] if isinstance(v, (list, tuple)) else v


def jasper_block_forward(x, lens, sd, i, lcfg):
    """JasperBlock.forward (parts/jasper.py:408-448) for the layouts the shipped configs use:
    residual_mode='add', no dense residual, no SE, groups=1, heads=-1, activation ReLU,
    dropout = identity (eval)."""
    k = _first(lcfg["kernel"])
    if k % 2 == 0:
        k += 1
    stride, dil = _first(lcfg["stride"]), _first(lcfg["dilation"])
    pad = get_same_padding(k, stride, dil)
    rep, sep = lcfg["repeat"], lcfg.get("separable", False)
    lens_orig, x_in = lens, x
    out, j = x, 0
    for r in range(rep):
        p = f"encoder.{i}.mconv"
        if sep:
            w = _t(sd[f"{p}.{j}.conv.weight"], x.dtype)
            out, lens = masked_conv1d(out, lens, w, stride, pad, dil, groups=w.shape[0])
            out, lens = masked_conv1d(out, lens, _t(sd[f"{p}.{j + 1}.conv.weight"], x.dtype))
            out = _bn_eval(out, sd, f"{p}.{j + 2}")
            j += 3
        else:
            out, lens = masked_conv1d(out, lens, _t(sd[f"{p}.{j}.conv.weight"], x.dtype), stride, pad, dil)
            out = _bn_eval(out, sd, f"{p}.{j + 1}")
            j += 2
        if r != rep - 1:
            out = F.relu(out)
            j += 2
    if lcfg["residual"]:
        p = f"encoder.{i}.res.0"
        res, _ = masked_conv1d(x_in, lens_orig, _t(sd[f"{p}.0.conv.weight"], x.dtype))
        res = _bn_eval(res, sd, f"{p}.1")
        out = out + res                                                        # :438-439
    return F.relu(out), lens                                                   # :444 mout


def encoder_forward(mel, length, sd, jasper_cfg, dtype=torch.float32):
    """JasperEncoder.forward (jasper.py:198-204): Sequential of JasperBlocks.
    Returns (outputs [B,C,T'] f32, encoded_lengths [B] float32 -- quirk Q3).
    dtype=torch.float64 runs the SAME graph in double precision: not the reference's arithmetic (that is float32, the
    default) but the reference's function without its rounding -- the tests use it to tell a frame on which two float32
    computations may legitimately disagree (a top-2 tie inside float32 rounding) from a wrong answer."""
    x = torch.as_tensor(mel).to(dtype)
    lens = torch.as_tensor(length)
    with torch.no_grad():
        for i, l in enumerate(jasper_cfg):
            x, lens = jasper_block_forward(x, lens, sd, i, l)
    return x, lens


# --------------------------------------------------------------------------- A9-A11
def decoder_forward(enc, sd):
    """JasperDecoderForCTC.forward (jasper.py:253-254): 1x1 conv + bias -> transpose -> log_softmax.  (Runs in the dtype of
    `enc`: float32 = the reference's arithmetic; float64 see encoder_forward.)"""
    with torch.no_grad():
        enc = torch.as_tensor(enc)
        y = F.conv1d(enc, _t(sd["decoder_layers.0.weight"], enc.dtype), _t(sd["decoder_layers.0.bias"], enc.dtype))
        return F.log_softmax(y.transpose(1, 2), dim=-1)


def greedy_argmax(log_probs):
    """GreedyCTCDecoder.forward (greedy_ctc_decoder.py:33-36); ties -> lowest index (Q6)."""
    return torch.as_tensor(log_probs).argmax(dim=-1, keepdim=False)


def ctc_collapse_ids(pred_row, blank_id):
    """Inner loop of __ctc_decoder_predictions_tensor (helpers.py:24-31): over ALL frames (Q4)."""
    out, previous = [], blank_id
    for p in pred_row:
        p = int(p)
        if (p != previous or previous == blank_id) and p != blank_id:
            out.append(p)
        previous = p
    return out


def ctc_decode_strings(predictions, labels):
    """post_process_predictions for one [B,T'] tensor (helpers.py:7-33, 207-208)."""
    blank = len(labels)
    return ["".join(labels[c] for c in ctc_collapse_ids(row, blank)) for row in np.asarray(predictions)]


# --------------------------------------------------------------------------- whole path
def forward_all(signal, length, enc_sd, dec_sd, jasper_cfg, fb=None, **pre):
    """infer.py:146-160 DAG with the greedy decoder (infer.py:113): returns a dict of every port tensor."""
    mel, seq = melspec_forward(signal, length, fb=fb, **pre)
    enc, enc_len = encoder_forward(mel, seq, enc_sd, jasper_cfg)
    logp = decoder_forward(enc, dec_sd)
    pred = greedy_argmax(logp)
    return dict(mel=mel, seq=seq, enc=enc, enc_len=enc_len, logp=logp, pred=pred)


def out_frames(samples, hop=160):
    """T = 1 + L//hop (torch.stft center=True), T' after the stride-2 block."""
    t = 1 + samples // hop
    return t, (t - 1) // 2 + 1
