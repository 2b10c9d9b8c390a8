"""CPU oracle for the resampler -- TEST INFRASTRUCTURE ONLY.

*** PARITY UNPINNED *** The reference resamples with ``librosa.load(sr=16000)`` (infer.py:200) whose default
``res_type='kaiser_best'`` is the third-party package resampy (absent here).  This restates resampy's published
``resample_f`` loop (interpolated windowed sinc) in numpy and librosa's length rule around it:

    librosa.resample(y, orig_sr, target_sr, fix=True):  ratio = float(target_sr) / orig_sr
        n_samples = int(np.ceil(y.shape[-1] * ratio));  y_hat = resampy.resample(...)   # int(n * ratio) samples
        y_hat = util.fix_length(y_hat, n_samples)                                        # zero-padded to ceil(n * ratio)

Two known differences from the packages, both inside the 2e-6 bound the device is held to: resampy accumulates in the
INPUT dtype (float32 for librosa.load's output) where this loop accumulates in float64, and its table is float64 where
the device's is float32.
"""
import numpy as np


def sinc_window(num_zeros=64, precision=9, rolloff=0.9475937167399596, beta=14.769656459379492):
    num_bits = 2 ** precision
    n = num_bits * num_zeros
    sinc_win = rolloff * np.sinc(rolloff * np.linspace(0, num_zeros, num=n + 1, endpoint=True))
    taper = np.kaiser(2 * n + 1, beta)[n:]
    return taper * sinc_win, num_bits


def resample(x, sr_orig, sr_new, **kw):
    x = np.asarray(x, dtype=np.float64)
    ratio = float(sr_new) / sr_orig
    n_out = int(x.shape[0] * ratio)                 # resampy: shape[axis] = int(shape[axis] * sample_ratio)
    n_fixed = int(np.ceil(x.shape[0] * ratio))      # librosa.resample(fix=True)
    interp_win, num_table = sinc_window(**kw)
    if ratio < 1:
        interp_win = interp_win * ratio
    interp_delta = np.zeros_like(interp_win)
    interp_delta[:-1] = np.diff(interp_win)
    scale = min(1.0, ratio)
    index_step = int(scale * num_table)
    nwin, n_orig = interp_win.shape[0], x.shape[0]
    y = np.zeros(n_fixed)
    for t in range(n_out):
        time_register = t / ratio
        n = int(time_register)
        frac = scale * (time_register - n)
        index_frac = frac * num_table
        offset = int(index_frac)
        eta = index_frac - offset
        i_max = min(n + 1, (nwin - offset) // index_step)
        idx = offset + np.arange(i_max) * index_step
        y[t] += np.dot(interp_win[idx] + eta * interp_delta[idx], x[n - np.arange(i_max)])
        frac = scale - frac
        index_frac = frac * num_table
        offset = int(index_frac)
        eta = index_frac - offset
        k_max = min(n_orig - n - 1, (nwin - offset) // index_step)
        idx = offset + np.arange(k_max) * index_step
        y[t] += np.dot(interp_win[idx] + eta * interp_delta[idx], x[n + 1 + np.arange(k_max)])
    return y.astype(np.float32)
