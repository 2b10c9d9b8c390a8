"""Batched manifest-driven data layer: counterpart of ``AudioToTextDataLayer`` for inference / evaluation.

Reference: nemo/collections/asr/data_layer.py:42-190 (ports, manifest arguments),
parts/manifest.py:21-94 (JSON-lines entries with ``audio_filepath``, ``duration``, ``text``),
parts/dataset.py:14-53 (``seq_collate_fn``: zero-pad signals and tokens to the batch maximum),
parts/parsers.py (character parser: unknown characters dropped).  Decoding is PCM-WAV only (audio.py); files at
another rate are resampled on the device by the caller (``VietASR`` / ``audio.resample``), so this layer tags
every batch with its source rate.  With ``AllGpu`` placement the utterance list is sharded by rank: by default in
duration-balanced length buckets (``dist.balanced_shards``; ``shard_by="count"`` keeps contiguous equal-count shards
in manifest order -- the reference shards with a DistributedSampler, data_layer.py:161-165, equal counts too).
"""
import json

import numpy as np
import torch

from .audio import read_wav
from .core import AudioSignal, DataLayerNM, DeviceType, LengthsType, NeuralType
from .dist import balanced_shards, shard_range


class ChannelIndexType(LengthsType):
    pass


class AudioToTextDataLayer(DataLayerNM):
    @property
    def output_ports(self):
        return {"audio_signal": NeuralType(("B", "T"), AudioSignal(freq=self._sample_rate)),
                "a_sig_length": NeuralType(tuple("B"), LengthsType()),
                "transcripts": NeuralType(("B", "T"), ChannelIndexType()),
                "transcript_length": NeuralType(tuple("B"), LengthsType())}

    def __init__(self, manifest_filepath, labels, batch_size, sample_rate=16000, min_duration=0.1, max_duration=None,
                 shuffle=False, bucket_by_length=True, drop_last=False, pad_id=None, shard_by="duration", **_unused):
        super().__init__()
        self._sample_rate, self._batch_size, self._shuffle = sample_rate, batch_size, shuffle
        self.labels = list(labels)
        self._lab = {c: i for i, c in enumerate(self.labels)}
        self.pad_id = len(self.labels) if pad_id is None else pad_id
        items = []
        for path in str(manifest_filepath).split(","):
            with open(path, encoding="utf-8") as f:
                for line in f:
                    if not line.strip():
                        continue
                    e = json.loads(line)
                    d = float(e.get("duration", 0.0))
                    if (min_duration and d and d < min_duration) or (max_duration and d > max_duration):
                        continue                       # manifest.py filters by duration the same way
                    items.append((e["audio_filepath"], d, e.get("text", "")))
        if shard_by not in ("duration", "count"):
            raise ValueError(f"shard_by must be 'duration' or 'count', got {shard_by!r}")
        self.manifest_index = list(range(len(items)))        # position of every kept item in the (filtered) manifest
        batches = None
        if self.placement == DeviceType.AllGpu and torch.distributed.is_available() and torch.distributed.is_initialized():
            rank, world = torch.distributed.get_rank(), torch.distributed.get_world_size()
            # duration-balanced dealing fixes the batch composition (length buckets), so it is only taken when the caller did
            # not ask for something it cannot honour: shuffle=True (the reference's DistributedSampler shuffles
            # utterances, actions.py:669-693) and bucket_by_length=False fall back to equal-count contiguous shards
            if shard_by == "duration" and all(it[1] > 0 for it in items) and bucket_by_length and not shuffle:
                # length buckets of batch_size utterances dealt heaviest-first to the least loaded rank: the ranks'
                # padded work (rows x longest row, summed over batches) ends up within a few percent of each other
                batches = [sorted(b, key=lambda i: (items[i][1], i)) for b in balanced_shards([it[1] for it in items], world, batch_size)[rank]]
                batches.sort(key=lambda b: items[b[0]][1])
            else:
                lo, hi = shard_range(len(items), rank, world)
                self.manifest_index = self.manifest_index[lo:hi]
                items = items[lo:hi]
        order = list(range(len(items)))
        if bucket_by_length and not shuffle:
            order.sort(key=lambda i: items[i][1])    # similar lengths share a batch: less padding work
        self._items, self._order = items, order
        self._batches = batches if batches is not None else [order[i:i + batch_size] for i in range(0, len(order), batch_size)]
        if drop_last:     # (the duration-balanced path sorts its batches by length: the short one is not the last)
            self._batches = [b for b in self._batches if len(b) == batch_size]

    def tokens(self, text):
        return [self._lab[c] for c in text if c in self._lab]

    def __len__(self):
        return len(self._batches)

    @property
    def dataset(self):
        return None

    @property
    def data_iterator(self):
        return _BatchIter(self)

    def utterance_order(self):
        """Indices into the (duration-filtered) manifest, in the order the batches deliver them."""
        return [self.manifest_index[i] for b in self._batches for i in b]


class _BatchIter:
    def __init__(self, layer):
        self.layer, self.i = layer, 0
        if layer._shuffle:
            np.random.shuffle(layer._batches)

    def __len__(self):
        return len(self.layer._batches)

    def __iter__(self):
        return self

    def __next__(self):
        L = self.layer
        if self.i >= len(L._batches):
            raise StopIteration
        idx = L._batches[self.i]
        self.i += 1
        sigs, toks = [], []
        for j in idx:
            path, _, text = L._items[j]
            x, sr = read_wav(path)
            if sr != L._sample_rate:
                raise ValueError(f"{path}: sample rate {sr} != {L._sample_rate}; resample first (audio.resample)")
            sigs.append(x)
            toks.append(L.tokens(text))
        a_len = torch.tensor([len(s) for s in sigs], dtype=torch.int64)
        audio = torch.zeros((len(sigs), int(a_len.max())), dtype=torch.float32)
        for k, s in enumerate(sigs):
            audio[k, : len(s)] = torch.from_numpy(s)
        t_len = torch.tensor([len(t) for t in toks], dtype=torch.int64)
        tokens = torch.full((len(toks), max(int(t_len.max()), 1)), L.pad_id, dtype=torch.int64)
        for k, t in enumerate(toks):
            tokens[k, : len(t)] = torch.tensor(t, dtype=torch.int64)
        return audio, a_len, tokens, t_len


def word_error_rate(hypotheses, references, use_cer=False):
    """nemo/collections/asr/metrics.py:30-63: summed Levenshtein distance / summed reference length."""
    def lev(a, b):
        prev = list(range(len(b) + 1))
        for i, x in enumerate(a, 1):
            cur = [i]
            for j, y in enumerate(b, 1):
                cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (x != y)))
            prev = cur
        return prev[-1]
    if len(hypotheses) != len(references):
        raise ValueError("In word error rate calculation, hypotheses and reference lists must have the same number of "
                         f"elements. But I got: {len(hypotheses)} and {len(references)} correspondingly")
    scores = words = 0
    for h, r in zip(hypotheses, references):
        h_list, r_list = (list(h), list(r)) if use_cer else (h.split(), r.split())
        words += len(r_list)
        scores += lev(h_list, r_list)
    return 1.0 * scores / words if words else float("inf")
