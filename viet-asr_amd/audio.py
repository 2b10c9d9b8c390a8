"""Audio ingest for the callers of the path: WAV decoding, int -> float scaling, device resampling.

Reference behaviour being mirrored:
  * ``AudioSegment`` (nemo/collections/asr/parts/segment.py:19-32, 61-74): integer samples scaled by
    2^-(bits-1) to float32, multi-channel averaged to mono, optional resample to the target rate;
  * the CLI / web app decode with ``librosa.load(path, sr=16000)`` (infer.py:200, app.py:66,82), whose default
    resampler is resampy's ``kaiser_best`` -- third-party and absent here (parity unpinned); ``sinc_table`` /
    ``resample`` implement that published scheme on the device (csrc/audio.hip).
File decoding uses the standard library ``wave`` module (PCM WAV only; the reference's soundfile/librosa stack is
not in this image).
"""
import wave

import numpy as np
import torch

from . import _lib

KAISER_BEST = dict(num_zeros=64, precision=9, rolloff=0.9475937167399596, beta=14.769656459379492)
KAISER_FAST = dict(num_zeros=16, precision=9, rolloff=0.85, beta=8.555504641634386)


def read_wav(path):
    """-> (samples float32 mono in [-1, 1), sample_rate).  8/16/24/32-bit PCM."""
    with wave.open(path, "rb") as w:
        sr, nch, width, n = w.getframerate(), w.getnchannels(), w.getsampwidth(), w.getnframes()
        raw = w.readframes(n)
    if width == 1:
        x = (np.frombuffer(raw, dtype=np.uint8).astype(np.float32) - 128.0) / 128.0
    elif width == 2:
        x = np.frombuffer(raw, dtype="<i2").astype(np.float32) * (1.0 / 2 ** 15)
    elif width == 3:
        b = np.frombuffer(raw, dtype=np.uint8).reshape(-1, 3).astype(np.int32)
        v = b[:, 0] | (b[:, 1] << 8) | (b[:, 2] << 16)
        v = np.where(v >= 1 << 23, v - (1 << 24), v)
        x = v.astype(np.float32) * (1.0 / 2 ** 23)
    elif width == 4:
        x = np.frombuffer(raw, dtype="<i4").astype(np.float32) * (1.0 / 2 ** 31)
    else:
        raise TypeError(f"Unsupported sample width: {width} bytes")
    if nch > 1:
        x = x.reshape(-1, nch).mean(axis=1)          # segment.py:31-32
    return x.astype(np.float32), sr


def write_wav(path, samples, sr):
    """float32 [-1,1) -> 16-bit PCM WAV (test / tooling helper)."""
    pcm = np.clip(np.round(np.asarray(samples, dtype=np.float64) * 32768.0), -32768, 32767).astype("<i2")
    with wave.open(path, "wb") as w:
        w.setnchannels(1)
        w.setsampwidth(2)
        w.setframerate(int(sr))
        w.writeframes(pcm.tobytes())


def sinc_table(ratio=1.0, num_zeros=64, precision=9, rolloff=0.9475937167399596, beta=14.769656459379492):
    """One-sided Kaiser-windowed sinc + forward differences, as [n+1, 2] float32 (resampy ``sinc_window``)."""
    num_bits = 2 ** precision
    n = num_bits * num_zeros
    sinc_win = rolloff * np.sinc(rolloff * np.linspace(0, num_zeros, num=n + 1, endpoint=True))
    taper = np.kaiser(2 * n + 1, beta)[n:]
    win = taper * sinc_win
    if ratio < 1:
        win = win * ratio
    delta = np.zeros_like(win)
    delta[:-1] = np.diff(win)
    return np.stack([win, delta], axis=1).astype(np.float32), num_bits


_tables = {}


def resample(signal, length, sr_in, sr_out, filter="kaiser_best"):
    """signal [B, L] float32 cuda (zero padded), length [B] int64 cuda -> (signal' [B, L'], length').

    Lengths follow ``librosa.load(sr=...)`` -> ``librosa.resample(fix=True)`` (infer.py:200): resampy computes
    ``int(n * ratio)`` samples, librosa pads them with zeros to ``ceil(n * ratio)`` (both in float64: 5 000 samples at
    11 025 -> 16 000 Hz come out as 7 257, the last one zero)."""
    if sr_in == sr_out:
        return signal, length
    if signal.device.type != "cuda":
        raise _lib.VasrError("viet-asr_amd kernels need HIP-resident tensors; there is no CPU fallback for this path")
    params = {"kaiser_best": KAISER_BEST, "kaiser_fast": KAISER_FAST}[filter]
    ratio = float(sr_out) / float(sr_in)
    key = (signal.device, filter, ratio < 1 and ratio)
    if key not in _tables:
        tab, num_table = sinc_table(ratio, **params)
        _tables[key] = (torch.from_numpy(tab).to(signal.device), num_table)
    tab, num_table = _tables[key]
    x = signal.to(torch.float32).contiguous()
    ln = length.to(torch.int64).contiguous()
    B, L = x.shape
    L_out = int(np.ceil(L * ratio))
    y = torch.empty((B, max(L_out, 1)), dtype=torch.float32, device=x.device)
    ln_out = torch.empty((B,), dtype=torch.int64, device=x.device)
    _lib.check(_lib.lib().vasr_resample_f32(x.data_ptr(), L, ln.data_ptr(), B, tab.data_ptr(), tab.shape[0], num_table,
                                            ratio, y.data_ptr(), y.shape[1], ln_out.data_ptr(),
                                            torch.cuda.current_stream().cuda_stream))
    return y, ln_out


def pcm16_to_float(pcm):
    """int16 cuda tensor -> float32 / 32768 on the device (halves the host->device bytes of a batch)."""
    if pcm.device.type != "cuda" or pcm.dtype != torch.int16:
        raise ValueError("pcm16_to_float expects an int16 cuda tensor")
    p = pcm.contiguous()
    out = torch.empty(p.shape, dtype=torch.float32, device=p.device)
    _lib.check(_lib.lib().vasr_pcm16_to_f32(p.data_ptr(), p.numel(), out.data_ptr(), torch.cuda.current_stream().cuda_stream))
    return out
