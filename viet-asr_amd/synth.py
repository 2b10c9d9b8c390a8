"""Deterministic synthetic inputs for the QuartzNet CTC path.

The trained encoder checkpoints are absent from the reference mount
(/root/reference/.MISSING_LARGE_BLOBS:2), so parity fixtures, smoke() and
bench.py all run on weights drawn from a per-key counter-based generator and
on seeded synthetic 16 kHz audio.  Everything here is numpy
only (legacy ``RandomState`` streams are frozen across numpy versions), so the
dev container and the GPU box produce bit-identical tensors.

Key naming follows the reference ``state_dict`` layout
(nemo/backends/pytorch/nm.py:92-103, nemo/collections/asr/parts/jasper.py:329-400):
``encoder.{i}.mconv.{j}.conv.weight`` / ``encoder.{i}.mconv.{j}.{weight,bias,
running_mean,running_var,num_batches_tracked}`` / ``encoder.{i}.res.0.0.conv.weight``
/ ``encoder.{i}.res.0.1.*`` and ``decoder_layers.0.{weight,bias}``.
"""
import zlib

import numpy as np


# Gains chosen so that activation RMS stays O(1..10) through all 18 blocks of QuartzNet15x5 with the
# random BN statistics below (measured: 0.85 -> 2.4 for 12x1, 0.85 -> 6 for 15x5).
G_MAIN = 1.19
G_RES = 0.7


def _rs(key, seed):
    return np.random.RandomState((zlib.crc32(key.encode("utf-8")) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)


def _conv_weight(key, seed, cout, cin_g, k, gain=1.0):
    # fan-in scaled uniform keeps activations O(1) through ~80 layers
    fan_in = cin_g * k
    bound = gain * np.sqrt(3.0 / fan_in)
    return _rs(key, seed).uniform(-bound, bound, size=(cout, cin_g, k)).astype(np.float32)


def _bn(prefix, seed, c, out):
    r = _rs(prefix, seed)
    out[prefix + ".weight"] = r.uniform(0.8, 1.2, size=c).astype(np.float32)
    out[prefix + ".bias"] = r.normal(0.0, 0.1, size=c).astype(np.float32)
    out[prefix + ".running_mean"] = r.normal(0.0, 0.1, size=c).astype(np.float32)
    out[prefix + ".running_var"] = r.uniform(0.5, 1.5, size=c).astype(np.float32)
    out[prefix + ".num_batches_tracked"] = np.array(1000, dtype=np.int64)


def kernel_of(lcfg):
    """Effective (odd) kernel width of a block (parts/jasper.py:52-57, kernel_size_factor=1)."""
    k = lcfg["kernel"]
    k = k[0] if isinstance(k, (list, tuple)) else k
    f = float(lcfg.get("kernel_size_factor", 1.0))
    k = max(int(k * f), 1)
    if k % 2 == 0:
        k += 1
    return k


def first(v):
    return v[0] if isinstance(v, (list, tuple)) else v


def encoder_state_dict(jasper_cfg, feat_in, seed=0):
    """numpy state_dict for JasperEncoder(jasper=jasper_cfg, feat_in=feat_in)."""
    sd = {}
    cin = feat_in
    for i, l in enumerate(jasper_cfg):
        cout, rep, k = l["filters"], l["repeat"], kernel_of(l)
        sep = l.get("separable", False)
        c = cin
        j = 0
        for r in range(rep):
            p = f"encoder.{i}.mconv"
            if sep:
                sd[f"{p}.{j}.conv.weight"] = _conv_weight(f"{p}.{j}.conv.weight", seed, c, 1, k, gain=G_MAIN)
                sd[f"{p}.{j + 1}.conv.weight"] = _conv_weight(f"{p}.{j + 1}.conv.weight", seed, cout, c, 1, gain=G_MAIN)
                _bn(f"{p}.{j + 2}", seed, cout, sd)
                j += 3
            else:
                sd[f"{p}.{j}.conv.weight"] = _conv_weight(f"{p}.{j}.conv.weight", seed, cout, c, k, gain=G_MAIN)
                _bn(f"{p}.{j + 1}", seed, cout, sd)
                j += 2
            if r != rep - 1:
                j += 2  # activation + dropout slots
            c = cout
        if l["residual"]:
            p = f"encoder.{i}.res.0"
            sd[f"{p}.0.conv.weight"] = _conv_weight(f"{p}.0.conv.weight", seed, cout, cin, 1, gain=G_RES)
            _bn(f"{p}.1", seed, cout, sd)
        cin = cout
    return sd


def decoder_state_dict(feat_in, num_classes_with_blank, seed=0):
    sd = {}
    # gain 2 on O(1..6) activations: peaky posteriors, greedy argmax margins far above fp32 round-off
    sd["decoder_layers.0.weight"] = _conv_weight("decoder_layers.0.weight", seed, num_classes_with_blank, feat_in, 1, gain=2.0)
    sd["decoder_layers.0.bias"] = _rs("decoder_layers.0.bias", seed).normal(0, 0.5, size=num_classes_with_blank).astype(np.float32)
    return sd


def audio_batch(batch, samples, seed=0, ragged=False, amp=0.1):
    """(signal [B,L] f32 zero-padded past each length, length [B] i64).

    Smoothed uniform noise (a 3-tap low-pass over U(-amp, amp)); with ``ragged`` the
    lengths are uniform in [L/2, L] and the longest row is forced to L, mirroring the
    zero-pad-to-max collate of parts/dataset.py:14-53.
    """
    r = np.random.RandomState(1234567 + seed)
    x = r.uniform(-amp, amp, size=(batch, samples + 2)).astype(np.float32)
    x = (0.25 * x[:, :-2] + 0.5 * x[:, 1:-1] + 0.25 * x[:, 2:]).astype(np.float32)
    # slow amplitude envelope so the per-feature statistics are not flat
    t = np.arange(samples, dtype=np.float32) / 16000.0
    env = (0.6 + 0.4 * np.sin(2 * np.pi * (0.7 + 0.1 * np.arange(batch)[:, None]) * t[None, :])).astype(np.float32)
    x = (x * env).astype(np.float32)
    if ragged:
        lens = r.randint(samples // 2, samples + 1, size=batch).astype(np.int64)
        lens[r.randint(0, batch)] = samples
    else:
        lens = np.full(batch, samples, dtype=np.int64)
    for b in range(batch):
        x[b, lens[b]:] = 0.0
    return x, lens
