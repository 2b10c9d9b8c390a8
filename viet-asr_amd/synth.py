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


def band_limit(x, band_hz, rate=16000, taps=255, beta=8.6):
    """Rows of ``x`` low-passed at ``band_hz`` by a Kaiser-windowed sinc (about -85 dB in the stop band), float64
    accumulation in a fixed order (no FFT: the result must be the same bits wherever the fixtures are replayed).
    This is what 8 kHz-sourced audio looks like after its conversion to the model's 16 kHz (the reference's stated
    domain, README.md:21): the upper third of the mel bins holds nothing but the filter's floor."""
    n = np.arange(taps, dtype=np.float64) - (taps - 1) / 2
    h = 2.0 * band_hz / rate * np.sinc(2.0 * band_hz / rate * n) * np.kaiser(taps, beta)
    h /= h.sum()
    out = np.empty_like(x)
    for b in range(x.shape[0]):
        out[b] = np.convolve(x[b].astype(np.float64), h, mode="same").astype(np.float32)
    return out


def audio_batch(batch, samples, seed=0, ragged=False, amp=0.1, band_hz=None):
    """(signal [B,L] f32 zero-padded past each length, length [B] i64).

    Smoothed uniform noise (a 3-tap low-pass over U(-amp, amp)); with ``ragged`` the
    lengths are uniform in [L/2, L] and the longest row is forced to L, mirroring the
    zero-pad-to-max collate of parts/dataset.py:14-53.  ``band_hz``: additionally
    low-passed there (``band_limit``) and brought back to the same peak level.
    """
    r = np.random.RandomState(1234567 + seed)
    x = r.uniform(-amp, amp, size=(batch, samples + 2)).astype(np.float32)
    x = (0.25 * x[:, :-2] + 0.5 * x[:, 1:-1] + 0.25 * x[:, 2:]).astype(np.float32)
    # slow amplitude envelope so the per-feature statistics are not flat
    t = np.arange(samples, dtype=np.float32) / 16000.0
    env = (0.6 + 0.4 * np.sin(2 * np.pi * (0.7 + 0.1 * np.arange(batch)[:, None]) * t[None, :])).astype(np.float32)
    x = (x * env).astype(np.float32)
    if band_hz:
        x = band_limit(x, float(band_hz))
        x = (x * np.float32(amp / max(float(np.abs(x).max()), 1e-9))).astype(np.float32)
    if ragged:
        lens = r.randint(samples // 2, samples + 1, size=batch).astype(np.int64)
        lens[r.randint(0, batch)] = samples
    else:
        lens = np.full(batch, samples, dtype=np.int64)
    for b in range(batch):
        x[b, lens[b]:] = 0.0
    return x, lens


def synthetic_arpa(path, labels, n_words=20000, n_bigrams=50000, n_trigrams=50000, seed=0):
    """Write a deterministic back-off 3-gram model in ARPA text at a REALISTIC table size (defaults: 120 003 n-grams;
    the reference's own LMs -- .MISSING_LARGE_BLOBS:4-7, 3/4/5-gram binaries -- are absent) and return
    {tuple(words): (log10 p, log10 backoff)}.  Words are random strings over the letters of ``labels`` (2-8
    characters, Zipf-like unigram scores), bigrams / trigrams random tuples of them with back-off weights on every
    context that has an extension -- not a normalised model, but every table lookup, back-off step and <unk> path of
    a real one is exercised at a real hash-table load."""
    r = np.random.RandomState(0x5EED + seed)
    letters = [c for c in labels if c.strip() and c.isalpha()]
    words = set()
    while len(words) < n_words:
        for n in r.randint(2, 9, size=n_words):
            words.add("".join(letters[i] for i in r.randint(0, len(letters), size=n)))
            if len(words) == n_words:
                break
    words = sorted(words)
    rank = r.permutation(n_words)
    ng = {("<s>",): (-99.0, -0.35), ("</s>",): (-1.6, 0.0), ("<unk>",): (-4.5, 0.0)}
    uni_p = -1.5 - 0.9 * np.log10(1.0 + rank)                      # Zipf-like
    uni_bo = -0.6 * r.rand(n_words)
    for w, p, bo in zip(words, uni_p, uni_bo):
        ng[(w,)] = (round(float(p), 6), round(float(bo), 6))
    pop = np.argsort(rank)                                          # frequent words first
    def pick(n):                                                    # frequent words appear in more n-grams
        return pop[np.minimum((r.rand(n) ** 2.5 * n_words).astype(np.int64), n_words - 1)]
    a, b = pick(n_bigrams), pick(n_bigrams)
    for i in range(n_bigrams):
        ng[(words[a[i]], words[b[i]])] = (round(float(-0.3 - 2.5 * r.rand()), 6), round(float(-0.5 * r.rand()), 6))
    big = [k for k in ng if len(k) == 2]
    pre = r.randint(0, len(big), size=n_trigrams)
    c = pick(n_trigrams)
    for i in range(n_trigrams):
        ng[big[pre[i]] + (words[c[i]],)] = (round(float(-0.2 - 2.0 * r.rand()), 6), 0.0)
    for i in range(min(2000, n_bigrams)):                           # sentence starts and ends
        ng[("<s>", words[a[i]])] = (round(float(-1.0 - 2.0 * r.rand()), 6), round(float(-0.4 * r.rand()), 6))
        ng[(words[b[i]], "</s>")] = (round(float(-0.8 - 1.5 * r.rand()), 6), 0.0)
    with open(path, "w", encoding="utf-8") as f:
        f.write("\\data\\\n")
        by_n = {n: sorted(k for k in ng if len(k) == n) for n in (1, 2, 3)}
        for n in (1, 2, 3):
            f.write(f"ngram {n}={len(by_n[n])}\n")
        for n in (1, 2, 3):
            f.write(f"\n\\{n}-grams:\n")
            for k in by_n[n]:
                p, bo = ng[k]
                # (</s> without a back-off field, as KenLM prints it: in the model, not in the unigram list pyctcdecode reads)
                f.write(f"{p:.6f}\t{' '.join(k)}" + (f"\t{bo:.6f}" if n < 3 and k != ("</s>",) else "") + "\n")
        f.write("\n\\end\\\n")
    return ng


def ctc_like_log_probs(batch, frames, labels, words, seed=0, blank_frac=0.55, noise=1.3):
    """[batch, frames, len(labels)+1] float32 log-probabilities shaped like the output of a CONVERGED CTC model (the
    random-weight models here emit near-deterministic posteriors, ~1.1 classes per frame above pyctcdecode's
    token_min_logp, which leave a beam search nothing to do): each row spells a sequence of ``words`` (separated by
    ' '), every character held for 1-3 frames between runs of blank frames; frames carry Gaussian logit noise and now
    and then a competing character, so a handful of classes clear token_min_logp = -5 and beams genuinely branch,
    merge and get re-ranked by the language model at word boundaries.  Blank is the last class."""
    r = np.random.RandomState(0xC7C + seed)
    V = len(labels)
    lab = {c: i for i, c in enumerate(labels)}
    out = np.empty((batch, frames, V + 1), dtype=np.float32)
    for b in range(batch):
        z = noise * r.randn(frames, V + 1).astype(np.float32)
        t = 0
        prev = None
        while t < frames:
            w = words[r.randint(0, len(words))] + " "
            for ch in w:
                if t >= frames:
                    break
                gap = r.geometric(1.0 - blank_frac) - 1 + (1 if ch == prev else 0)     # a repeat needs a blank between
                z[t:t + gap, V] += 9.0
                t += gap
                hold = r.randint(1, 4)
                z[t:t + hold, lab[ch]] += 7.0 + 2.0 * r.rand()
                if r.rand() < 0.25 and t < frames:                                      # a competitor on the first frame
                    z[t, r.randint(0, V)] += 6.0
                t += hold
                prev = ch
        z = z.astype(np.float64)
        out[b] = (z - np.log(np.exp(z).sum(-1, keepdims=True))).astype(np.float32)
    return out
