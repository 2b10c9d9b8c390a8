"""world_size-2 gloo tests (CPU) of the multi-GPU result gather: dist.gather_id_sequences and the
executor's all_gather(shape) -> pad -> all_gather -> de-pad branch (actions.py:774-807 semantics)."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    import viet_asr_amd  # noqa: F401
    from viet_asr_amd import dist as vdist
    from viet_asr_amd.core import DataLayerNM, DeviceType, NeuralModuleFactory, NeuralType, NonTrainableNM
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # -- shard bookkeeping
        spans = [vdist.shard_range(7, r, world) for r in range(world)]
        assert spans == [(0, 4), (4, 7)]
        # -- ragged gather: rank 0 has 3 rows of width 5, rank 1 has 2 rows of width 8
        b, t = (3, 5) if rank == 0 else (2, 8)
        ids = (torch.arange(b * t, dtype=torch.int32).reshape(b, t) + 100 * rank)
        n = torch.tensor([t - i for i in range(b)], dtype=torch.int32)
        all_ids, all_n = vdist.gather_id_sequences(ids, n)
        assert all_ids.shape == (5, 8) and all_n.tolist() == [5, 4, 3, 8, 7]
        assert all_ids[0, :5].tolist() == [0, 1, 2, 3, 4] and all_ids[3, :8].tolist() == list(range(100, 108))

        # -- executor gather on a toy DAG (CPU tensors over gloo)
        class DL(DataLayerNM):
            output_ports = property(lambda self: {"x": NeuralType(("B", "T"))})

            def __init__(self):
                super().__init__()
                self._placement, self._device = DeviceType.AllGpu, torch.device("cpu")

            def __len__(self):
                return 1
            dataset = property(lambda self: None)

            @property
            def data_iterator(self):
                # rank 0 has TWO batches, rank 1 one: the executor must not hang in all_gather (unequal batch counts)
                class It(list):
                    pass
                return It([(torch.full((2 + rank, 3 + 2 * rank), float(rank)),)] +
                          ([(torch.full((1, 4), 7.0),)] if rank == 0 else []))

        class Twice(NonTrainableNM):
            input_ports = property(lambda self: {"x": NeuralType(("B", "T"))})
            output_ports = property(lambda self: {"y": NeuralType(("B", "T"))})

            def forward(self, x):
                return 2 * x + 1

        nf = NeuralModuleFactory(placement=DeviceType.CPU)
        y = Twice()(x=DL()())
        out = nf.infer([y])
        if rank == 0:
            parts = out[0]
            # de-padded per-rank shapes; the second step has a part from rank 0 only
            assert [tuple(p.shape) for p in parts] == [(2, 3), (3, 5), (1, 4)]
            assert parts[0].eq(1).all() and parts[1].eq(3).all() and parts[2].eq(15).all()
        else:
            assert out is None                                               # only rank 0 keeps results

        # -- dist.transcribe_sharded with a stand-in engine (same interface as engine.QuartzNetCTC): 5 utterances over
        #    2 ranks -> shards of 3 and 2 with different padded widths; every rank gets all transcripts in order
        class FakeEngine:
            device = torch.device("cpu")
            labels = list("abcdefghij")

            def forward(self, wav, length):
                ids = torch.zeros((wav.shape[0], wav.shape[1]), dtype=torch.int32)
                n = torch.zeros((wav.shape[0],), dtype=torch.int32)
                for b in range(wav.shape[0]):          # "transcript" = the first length[b] % 7 + 1 samples as label ids
                    k = int(length[b]) % 7 + 1
                    ids[b, :k] = wav[b, :k].to(torch.int32)
                    n[b] = k
                return dict(ids=ids, id_len=n)

            def texts(self, ids, n):
                return ["".join(self.labels[c] for c in ids[b, : n[b]].tolist()) for b in range(ids.shape[0])]

        import numpy as np
        sigs = [np.arange(L, dtype=np.float32) % 10 for L in (9, 15, 11, 30, 8)]
        want = ["".join("abcdefghij"[int(v)] for v in s[: len(s) % 7 + 1]) for s in sigs]
        assert vdist.transcribe_sharded(FakeEngine(), sigs) == want
        assert vdist.transcribe_sharded(FakeEngine(), sigs[:1]) == want[:1]      # fewer utterances than ranks
        q.put((rank, "ok"))
    except Exception as e:  # noqa: BLE001
        q.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_gather_world_size_2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=100) for _ in procs)
    for p in procs:
        p.join(30)
    assert res == {0: "ok", 1: "ok"}, res
