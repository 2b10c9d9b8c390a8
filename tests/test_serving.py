"""Dynamic-batching front end (viet-asr_amd/serving.py): host logic on CPU, end-to-end equality on the GPU."""
import threading
import time

import numpy as np
import pytest

import viet_asr_amd  # noqa: F401
from viet_asr_amd.serving import BatchingTranscriber


def _fake(calls):
    def fn(signals):
        calls.append([len(s) for s in signals])
        time.sleep(0.002)
        return [f"{len(s)}:{float(s[0]):.1f}" for s in signals]
    return fn


def test_plan_exact_groups_only_equal_lengths():
    srv = BatchingTranscriber(lambda s: [""] * len(s), max_batch=3)
    try:
        groups = srv.plan([5, 7, 5, 5, 7, 5, 9])
        assert sorted(map(sorted, groups)) == sorted([[0, 2, 3], [5], [1, 4], [6]])
    finally:
        srv.close()


def test_plan_padded_respects_ratio_and_max_batch():
    srv = BatchingTranscriber(lambda s: [""] * len(s), max_batch=4, policy="padded", max_pad_ratio=1.25)
    try:
        lens = [100, 110, 124, 126, 300, 101, 102]
        groups = srv.plan(lens)
        for g in groups:
            assert len(g) <= 4
            assert max(lens[i] for i in g) <= 1.25 * min(lens[i] for i in g)
        assert sorted(i for g in groups for i in g) == list(range(len(lens)))
    finally:
        srv.close()


def test_concurrent_requests_are_merged_and_answered_in_place():
    calls = []
    with BatchingTranscriber(_fake(calls), max_batch=8, max_wait_ms=50.0) as srv:
        sigs = [np.full(16, float(i), dtype=np.float32) for i in range(8)]
        out = [None] * 8

        def client(i):
            out[i] = srv.transcribe(sigs[i], timeout=10)
        ts = [threading.Thread(target=client, args=(i,)) for i in range(8)]
        [t.start() for t in ts]
        [t.join() for t in ts]
    assert out == [f"16:{float(i):.1f}" for i in range(8)]
    assert sum(len(c) for c in calls) == 8 and len(calls) < 8          # at least one merge happened
    assert srv.stats["requests"] == 8


def test_exact_policy_never_mixes_lengths_padded_does():
    calls = []
    with BatchingTranscriber(_fake(calls), max_batch=8, max_wait_ms=30.0) as srv:
        futs = [srv.submit(np.ones(n, dtype=np.float32)) for n in (10, 12, 10, 12, 10)]
        assert [f.result(10) for f in futs] == ["10:1.0", "12:1.0", "10:1.0", "12:1.0", "10:1.0"]
    assert all(len(set(c)) == 1 for c in calls)
    calls2 = []
    with BatchingTranscriber(_fake(calls2), max_batch=8, max_wait_ms=30.0, policy="padded", max_pad_ratio=1.5) as srv:
        futs = [srv.submit(np.ones(n, dtype=np.float32)) for n in (10, 12, 10, 12, 10)]
        [f.result(10) for f in futs]
    assert any(len(set(c)) > 1 for c in calls2)


def test_failure_reaches_every_waiting_request_and_server_survives():
    state = {"n": 0}

    def fn(signals):
        state["n"] += 1
        if state["n"] == 1:
            raise ValueError("device fell over")
        return ["ok"] * len(signals)
    with BatchingTranscriber(fn, max_batch=4, max_wait_ms=20.0) as srv:
        futs = [srv.submit(np.ones(4, dtype=np.float32)) for _ in range(3)]
        for f in futs:
            with pytest.raises(ValueError):
                f.result(10)
        assert srv.transcribe(np.ones(4, dtype=np.float32), timeout=10) == "ok"


def test_argument_and_lifecycle_errors():
    with pytest.raises(ValueError):
        BatchingTranscriber(lambda s: s, policy="greedy")
    with pytest.raises(ValueError):
        BatchingTranscriber(lambda s: s, max_batch=0)
    srv = BatchingTranscriber(lambda s: [""] * len(s))
    with pytest.raises(ValueError):
        srv.submit(np.zeros((2, 2), dtype=np.float32))
    srv.close()
    with pytest.raises(RuntimeError):
        srv.submit(np.ones(3, dtype=np.float32))


@pytest.mark.gpu
def test_served_answers_equal_unbatched_transcribe_on_device():
    import torch
    from viet_asr_amd import configs, synth
    from viet_asr_amd.engine import QuartzNetCTC
    assert torch.cuda.is_available()
    cfg = configs.builtin("quartznet12x1_vi")
    jas = cfg["JasperEncoder"]["jasper"]
    eng = QuartzNetCTC(cfg, synth.encoder_state_dict(jas, 64, 5), synth.decoder_state_dict(1024, len(cfg["labels"]) + 1, 5))
    rng = np.random.default_rng(5)
    lens = [16000, 24000, 16000, 16000, 24000, 31999]
    sigs = [(0.1 * rng.standard_normal(n)).astype(np.float32) for n in lens]
    alone = [eng.transcribe([s])[0] for s in sigs]
    with BatchingTranscriber(eng.transcribe, max_batch=8, max_wait_ms=200.0) as srv:
        futs = [srv.submit(s) for s in sigs]
        served = [f.result(120) for f in futs]
    assert served == alone
    assert max(srv.stats["device_calls_by_size"]) >= 2      # the equal-length requests went out together
