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


def gather_id_sequences(ids, id_len, group=None):
    """ids [B_loc, T] int32 (compacted rows), id_len [B_loc] int32 -> on every rank the concatenation over
    ranks in rank order as (ids [B_tot, T_max], id_len [B_tot]).  Shards may differ in B_loc and T."""
    world = dist.get_world_size(group)
    if world == 1:
        return ids, id_len
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
    return (torch.cat([a[:r] for a, r in zip(all_ids, rows)]), torch.cat([l[:r] for l, r in zip(all_len, rows)]))


def transcribe_sharded(engine, signals, group=None):
    """Each rank transcribes its contiguous shard of ``signals`` with the fused engine; every rank
    returns the full list of transcripts in the original order."""
    import numpy as np
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    lo, hi = shard_range(len(signals), rank, world)
    mine = signals[lo:hi]
    if mine:
        lens = np.array([len(s) for s in mine], dtype=np.int64)
        batch = np.zeros((len(mine), int(lens.max())), dtype=np.float32)
        for i, s in enumerate(mine):
            batch[i, : len(s)] = s
        r = engine.forward(torch.from_numpy(batch).to(engine.device), torch.from_numpy(lens).to(engine.device))
        ids, n = r["ids"], r["id_len"]
    else:
        ids = torch.zeros((0, 1), dtype=torch.int32, device=engine.device)
        n = torch.zeros((0,), dtype=torch.int32, device=engine.device)
    ids, n = gather_id_sequences(ids, n, group)
    return engine.texts(ids, n)
