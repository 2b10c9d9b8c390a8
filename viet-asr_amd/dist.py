"""Utterance sharding across the GPUs of one node (one process per GPU, RCCL over xGMI).

Utterances are independent, so the only communication on the path is the gather of results --
what the reference's ``PtActions._infer`` does for every returned tensor
(nemo/backends/pytorch/actions.py:774-807: all_gather(shape) -> pad to max -> all_gather(padded) ->
de-pad).  Here the collapsed id sequences are gathered with exactly two collectives per batch
(lengths + padded ids); messages are KB-sized, so latency, not link bandwidth, is what matters.
Backend "nccl" is RCCL on ROCm; the CPU tests run the same code over gloo.
"""
import torch
import torch.distributed as dist


def shard_range(n_items, rank, world_size):
    """Contiguous, balanced shard [lo, hi) of ``n_items`` utterances for ``rank``."""
    base, rem = divmod(n_items, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def balanced_shards(durations, world_size, batch_size=None):
    """Duration-balanced utterance shards: -> ``world_size`` lists of BATCHES (each a list of utterance indices).

    The cost of a padded batch is rows x longest row (every row is computed at the batch's width), so utterances are
    first bucketed by length -- sorted by duration, cut into batches of ``batch_size`` (None: one utterance per
    "batch") -- and the batches are then dealt to the ranks longest-processing-time-first: heaviest batch to the least
    loaded rank.  Contiguous count-based shards of a manifest in recording order can differ by the ratio of the longest
    to the shortest recordings; this keeps the ranks' padded work within a few percent ("length-bucketing
    matters more than the collective").  The reference shards with a DistributedSampler
    (nemo/backends/pytorch/actions.py:669-693): equal COUNTS per rank, no notion of length.
    Deterministic: ties go to the lower index / lower rank, so every rank computes the same assignment locally."""
    import heapq
    d = [float(x) for x in durations]
    order = sorted(range(len(d)), key=lambda i: (-d[i], i))
    bs = max(1, int(batch_size)) if batch_size else 1
    batches = [order[i:i + bs] for i in range(0, len(order), bs)]
    cost = [len(b) * d[b[0]] for b in batches]                  # b[0] is the longest row of its batch
    heap = [(0.0, r) for r in range(world_size)]
    shards = [[] for _ in range(world_size)]
    for k in sorted(range(len(batches)), key=lambda k: (-cost[k], k)):
        load, r = heapq.heappop(heap)
        shards[r].append(batches[k])
        heapq.heappush(heap, (load + cost[k], r))
    return shards


def shard_cost(shard, durations):
    """Padded work of one rank's list of batches: sum of rows x longest row."""
    return sum(len(b) * max(float(durations[i]) for i in b) for b in shard if b)


def job_passes(durations, rank, world_size, batch, bucket=64):
    """The passes ONE rank runs over its share of a ragged job (bench.py --config 5 --ragged; a manifest walked once):
    length buckets of ``bucket`` clips dealt to the ranks by ``balanced_shards``, this rank's buckets taken longest first and
    cut into passes of up to ``batch`` clips (similar lengths share a pass).  -> (passes: lists of manifest indices, padded
    work of every rank).  Ranks end up with DIFFERENT numbers of passes of different shapes, which is why the job's results
    go through ``gather_id_sequences`` (one shape-exchanging gather per step) and not through the fixed-shape ring."""
    shards = balanced_shards(durations, world_size, bucket)
    costs = [shard_cost(s, durations) for s in shards]
    mine = sorted((b for b in shards[rank]), key=lambda b: -float(durations[b[0]]))
    flat = [i for b in mine for i in b]
    batch = max(1, int(batch))
    return [flat[i:i + batch] for i in range(0, len(flat), batch)], costs


def gather_id_sequences(ids, id_len, group=None, extra=None):
    """ids [B_loc, T] int32 (compacted rows), id_len [B_loc] int32 -> on every rank the concatenation over
    ranks in rank order as (ids [B_tot, T_max], id_len [B_tot]).  Shards may differ in B_loc and T.
    extra: optional [B_loc] integer tensor gathered alongside (e.g. the rows' manifest indices); returned third."""
    world = dist.get_world_size(group)
    if world == 1:
        return (ids, id_len) if extra is None else (ids, id_len, extra)
    shape = torch.tensor(ids.shape, dtype=torch.int64, device=ids.device)
    shapes = [torch.empty_like(shape) for _ in range(world)]
    dist.all_gather(shapes, shape, group=group)
    mx = torch.stack(shapes).max(dim=0).values.tolist()
    padded = ids.new_zeros((mx[0], mx[1]))
    padded[: ids.shape[0], : ids.shape[1]] = ids
    plen = id_len.new_zeros((mx[0],))
    plen[: id_len.shape[0]] = id_len
    all_ids = [torch.empty_like(padded) for _ in range(world)]
    all_len = [torch.empty_like(plen) for _ in range(world)]
    dist.all_gather(all_ids, padded, group=group)
    dist.all_gather(all_len, plen, group=group)
    rows = [int(s[0]) for s in shapes]
    out = (torch.cat([a[:r] for a, r in zip(all_ids, rows)]), torch.cat([l[:r] for l, r in zip(all_len, rows)]))
    if extra is None:
        return out
    pex = extra.new_zeros((mx[0],))
    pex[: extra.shape[0]] = extra
    all_ex = [torch.empty_like(pex) for _ in range(world)]
    dist.all_gather(all_ex, pex, group=group)
    return out + (torch.cat([e[:r] for e, r in zip(all_ex, rows)]),)


class AsyncIdGather:
    """Double-buffered asynchronous gather of one batch's (ids, id_len) per step -- the steady-state form of the result
    gather for fixed-shape batches (every rank produces the same [B, T'] per step: the weak-scaling bench loop, a
    server's fixed batch).  One ``all_gather_into_tensor`` per returned tensor, like actions.py:774-807 issues one
    collective per tensor, but asynchronously on the backend's own stream into one of ``slots`` buffer pairs: the next
    batch's kernels do not wait for the other ranks, a buffer pair is reused only after its previous gather has been
    waited for, ``drain()`` waits for everything in flight.  The source tensors of a submitted gather are kept alive
    until it has been drained.  (Shapes that differ between ranks go through ``gather_id_sequences``.)"""

    def __init__(self, world, device, group=None, slots=2):
        self.world, self.device, self.group, self.slots = world, device, group, slots
        self._bufs, self._inflight, self._n, self._last = {}, [None] * slots, 0, None
        self.max_shapes = 4      # buffer sets kept: a loop that drifts through many [B, T'] shapes must not grow without bound

    def drain(self, slot=None):
        for k in (range(self.slots) if slot is None else (slot,)):
            if self._inflight[k] is not None:
                for w in self._inflight[k][:2]:
                    w.wait()
                self._inflight[k] = None

    def submit(self, ids, id_len):
        """Enqueue the gather of this rank's batch; returns the slot it went into."""
        shape = tuple(ids.shape)
        if shape not in self._bufs:
            # drop the buffer sets of shapes that have nothing in flight any more, oldest first (dict order = insertion order)
            busy = {f[4] for f in self._inflight if f is not None}
            for old in [k for k in self._bufs if k not in busy and (self._last is None or k != self._last[0])]:
                if len(self._bufs) < self.max_shapes:
                    break
                del self._bufs[old]
            # (the rank-major concatenation along dim 0: the output form both RCCL and gloo accept; read as [world, B, ...])
            self._bufs[shape] = [(torch.empty((self.world * shape[0],) + shape[1:], dtype=ids.dtype, device=self.device).view((self.world,) + shape),
                                  torch.empty((self.world * shape[0],), dtype=id_len.dtype, device=self.device).view(self.world, shape[0]))
                                 for _ in range(self.slots)]
        slot = self._n % self.slots
        self._n += 1
        self.drain(slot)
        g_ids, g_len = self._bufs[shape][slot]
        self._inflight[slot] = (dist.all_gather_into_tensor(g_ids.view((-1,) + shape[1:]), ids, group=self.group, async_op=True),
                                dist.all_gather_into_tensor(g_len.view(-1), id_len, group=self.group, async_op=True), ids, id_len, shape)
        self._last = (shape, slot)
        return slot

    def last(self):
        """(ids [world, B, T'], id_len [world, B]) of the most recent submit, waited for."""
        if self._last is None:
            raise RuntimeError("AsyncIdGather.last() before the first submit()")
        shape, slot = self._last
        self.drain(slot)
        return self._bufs[shape][slot]


def transcribe_sharded(engine, signals, group=None, balance=True, batch_size=None):
    """Each rank transcribes its shard of ``signals`` with the fused engine; every rank returns the full list of
    transcripts in the original order.  balance=True (default): duration-balanced shards (``balanced_shards``: length
    buckets of ``batch_size`` utterances, dealt heaviest-first to the least loaded rank; batch_size=None = the whole
    shard as one padded batch per rank, utterances dealt one by one); balance=False: contiguous shards by count, in
    input order, one batch per rank (round 1-2 behaviour)."""
    import numpy as np
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    if balance:
        mine = balanced_shards([len(s) for s in signals], world, batch_size)[rank]
        if batch_size is None:                       # one padded batch per rank
            mine = [[i for b in mine for i in b]] if mine else []
    else:
        lo, hi = shard_range(len(signals), rank, world)
        mine = [list(range(lo, hi))] if hi > lo else []
    parts_ids, parts_n, parts_idx = [], [], []
    for idx in mine:
        lens = np.array([len(signals[i]) for i in idx], dtype=np.int64)
        batch = np.zeros((len(idx), int(lens.max())), dtype=np.float32)
        for k, i in enumerate(idx):
            batch[k, : lens[k]] = signals[i]
        r = engine.forward(torch.from_numpy(batch).to(engine.device), torch.from_numpy(lens).to(engine.device))
        parts_ids.append(r["ids"]); parts_n.append(r["id_len"])
        parts_idx.append(torch.tensor(idx, dtype=torch.int32, device=engine.device))
    if parts_ids:
        width = max(p.shape[1] for p in parts_ids)
        ids = torch.cat([torch.nn.functional.pad(p, (0, width - p.shape[1])) for p in parts_ids])
        n, idx_t = torch.cat(parts_n), torch.cat(parts_idx)
    else:
        ids = torch.zeros((0, 1), dtype=torch.int32, device=engine.device)
        n = torch.zeros((0,), dtype=torch.int32, device=engine.device)
        idx_t = torch.zeros((0,), dtype=torch.int32, device=engine.device)
    ids, n, idx_t = gather_id_sequences(ids, n, group, extra=idx_t)
    texts = engine.texts(ids, n)
    out = [None] * len(signals)
    for t, i in zip(texts, idx_t.tolist()):
        out[i] = t
    return out
