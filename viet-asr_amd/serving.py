"""In-process serving front end with dynamic batching.

The reference serves one utterance at a time: ``app.py:58-89`` calls ``vietasr.transcribe(audio_signal)`` from each
request handler against one global model (``app.py:22-28``), batch size 1, no locking.  On an MI355X a batch of 64
costs 3.5x the time of a batch of 1, so concurrent requests are worth merging.  This module is the
queue between request threads and the device: no sockets, no HTTP -- the web layer stays out of scope.

    server = BatchingTranscriber(vietasr.transcribe_batch, max_batch=64, max_wait_ms=4.0)
    fut = server.submit(signal)          # from any thread; returns concurrent.futures.Future[str]
    text = server.transcribe(signal)     # blocking convenience, same call shape as VietASR.transcribe
    server.close()

Batching policy.  Results of the reference depend on the padded batch an utterance sits in (SURVEY §8 quirks Q4/Q5:
padded frames are decoded, reflect padding sees the padded row), and the reference always runs B = 1.  So:
  * ``policy="exact"`` (default) only merges requests with the same number of samples: every answer is identical to
    what ``transcribe`` returns for that signal alone;
  * ``policy="padded"`` merges anything up to ``max_batch`` / ``max_pad_ratio`` (zero-pad-to-max collate,
    parts/dataset.py:14-53): the reference's *batched* semantics, maximum throughput;
  * ``policy="independent"`` merges like "padded" but calls the engine with ``row_independent=True`` (include/vasr.h,
    vasr_set_row_independent: every row reflects at its own end and is decoded over its own frames): every answer is
    identical to what ``transcribe`` returns for that signal alone, at the throughput of "padded".  Needs an engine
    callable that takes the keyword (``VietASR.launch_batch`` / ``transcribe_batch`` do).
Requests are taken in arrival order; a cycle waits at most ``max_wait_ms`` after its first request.

Pipelining.  With ``launch_batch=vietasr.launch_batch`` (returns at once with a handle whose ``.texts()`` waits) the
worker collates and enqueues the next group while the device still runs the previous one, and a second thread
completes the futures: the host side of a batch (collate into pinned memory, PCIe, ids -> str) overlaps the kernels
instead of adding to them.  ``tools/bench_serving.py`` measures both.
"""
import queue
import threading
import time
from concurrent.futures import Future

import numpy as np


class BatchingTranscriber:
    def __init__(self, transcribe_batch=None, max_batch=64, max_wait_ms=4.0, policy="exact", max_pad_ratio=1.25,
                 launch_batch=None):
        if policy not in ("exact", "padded", "independent"):
            raise ValueError(f"policy must be 'exact', 'padded' or 'independent', got {policy!r}")
        if max_batch < 1:
            raise ValueError("max_batch must be >= 1")
        if (transcribe_batch is None) == (launch_batch is None):
            raise ValueError("give exactly one of transcribe_batch and launch_batch")
        kw = {"row_independent": True} if policy == "independent" else {}
        self._fn = (lambda sigs: transcribe_batch(sigs, **kw)) if transcribe_batch is not None else None
        self._launch = (lambda sigs: launch_batch(sigs, **kw)) if launch_batch is not None else None
        self.max_batch = int(max_batch)
        self.max_wait = float(max_wait_ms) / 1e3
        self.policy = policy
        self.max_pad_ratio = float(max_pad_ratio)
        self._q = queue.Queue()
        self._closed = False
        self.stats = {"requests": 0, "batches": 0, "device_calls_by_size": {}}
        self._inflight = queue.Queue(maxsize=2)      # (futures, pending handle): the engine holds two batches
        self._finisher = None
        if launch_batch is not None:
            self._finisher = threading.Thread(target=self._finish, name="vasr-finisher", daemon=True)
            self._finisher.start()
        self._worker = threading.Thread(target=self._run, name="vasr-batcher", daemon=True)
        self._worker.start()

    # ---- client side ----
    def submit(self, audio_signal):
        if self._closed:
            raise RuntimeError("BatchingTranscriber is closed")
        sig = np.asarray(audio_signal)
        sig = np.ascontiguousarray(sig, dtype=np.int16 if sig.dtype == np.int16 else np.float32)
        if sig.ndim != 1 or sig.size == 0:
            raise ValueError("audio_signal must be a non-empty 1-D array")
        fut = Future()
        self._q.put((sig, fut))
        return fut

    def transcribe(self, audio_signal, timeout=None):
        return self.submit(audio_signal).result(timeout)

    def close(self):
        if not self._closed:
            self._closed = True
            self._q.put(None)
            self._worker.join()
            if self._finisher is not None:
                self._inflight.put(None)
                self._finisher.join()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # ---- worker ----
    def _collect(self):
        """First request (blocking), then whatever arrives within max_wait, up to max_batch -- and, without waiting any
        longer, whatever else is already queued (up to 8 batches' worth): under load plan() then has enough requests
        of similar length to fill its groups."""
        first = self._q.get()
        if first is None:
            return None
        items = [first]
        deadline = time.monotonic() + self.max_wait
        while len(items) < 8 * self.max_batch:
            left = deadline - time.monotonic() if len(items) < self.max_batch else 0.0
            try:
                nxt = self._q.get(timeout=left) if left > 0 else self._q.get_nowait()
            except queue.Empty:
                break
            if nxt is None:              # close(): finish what is queued, then stop
                self._q.put(None)
                break
            items.append(nxt)
        return items

    def plan(self, lengths):
        """Index groups that go to the device together (pure function of the lengths; unit-tested on CPU)."""
        order = sorted(range(len(lengths)), key=lambda i: (lengths[i], i))
        groups, cur = [], []
        for i in order:
            if cur:
                same = lengths[i] == lengths[cur[0]]
                fits = same if self.policy == "exact" else lengths[i] <= self.max_pad_ratio * lengths[cur[0]]
                if not fits or len(cur) == self.max_batch:
                    groups.append(cur)
                    cur = []
            cur.append(i)
        if cur:
            groups.append(cur)
        return groups

    def _resolve(self, futs, get_texts):
        try:
            texts = get_texts()
            if len(texts) != len(futs):
                raise RuntimeError(f"the engine returned {len(texts)} results for {len(futs)} signals")
        except BaseException as e:   # noqa: BLE001 -- the request threads must see the failure
            for f in futs:
                f.set_exception(e)
            return
        by = self.stats["device_calls_by_size"]
        by[len(futs)] = by.get(len(futs), 0) + 1
        for f, t in zip(futs, texts):
            f.set_result(t)

    def _finish(self):
        while True:
            job = self._inflight.get()
            if job is None:
                return
            self._resolve(job[0], job[1].texts)

    def _run(self):
        while True:
            items = self._collect()
            if items is None:
                return
            self.stats["requests"] += len(items)
            self.stats["batches"] += 1
            for grp in self.plan([len(s) for s, _ in items]):
                futs = [items[i][1] for i in grp]
                sigs = [items[i][0] for i in grp]
                if self._launch is None:
                    self._resolve(futs, lambda: self._fn(sigs))
                    continue
                try:
                    pending = self._launch(sigs)
                except BaseException as e:   # noqa: BLE001
                    for f in futs:
                        f.set_exception(e)
                    continue
                self._inflight.put((futs, pending))
