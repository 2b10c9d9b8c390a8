#!/usr/bin/env python3
"""Concurrency stress of the serving queue (dev tool, GPU): python tests/devtools/stress_serving.py [seconds] [threads] [seed]

`threads` client threads submit requests of random length (a pool of 300 distinct signals, float32 or int16 PCM, 0.3-8 s) to ONE
BatchingTranscriber in the pipelined, row-independent configuration (launch_batch = engine.launch, policy "independent"): two
batches in flight on two staging slots, a worker thread collating, a finisher thread completing futures, the copy stream and the
compute stream overlapping.  Every answer must be the transcript the engine returns for that signal ALONE (computed up front, one
blocking call each).  A race in the staging slots, the per-stream workspaces or the event hand-over shows as a wrong or crossed
answer.  Prints one JSON line."""
import json
import os
import sys
import threading
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import viet_asr_amd  # noqa: E402,F401
from viet_asr_amd import configs, synth  # noqa: E402
from viet_asr_amd.engine import QuartzNetCTC  # noqa: E402
from viet_asr_amd.serving import BatchingTranscriber  # noqa: E402


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    n_threads = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    seed = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    cfg = configs.builtin("quartznet12x1_vi")
    jas = cfg["JasperEncoder"]["jasper"]
    eng = QuartzNetCTC(cfg, synth.encoder_state_dict(jas, 64, 5), synth.decoder_state_dict(1024, len(cfg["labels"]) + 1, 5))
    rng = np.random.default_rng(seed)
    pool = []
    for i in range(300):
        n = int(rng.integers(4800, 128000))
        if i % 3 == 0:
            pool.append(rng.integers(-6000, 6000, size=n).astype(np.int16))
        else:
            pool.append((float(rng.choice([0.02, 0.2])) * rng.standard_normal(n)).astype(np.float32))
    alone = [eng.transcribe([s], True)[0] for s in pool]          # row_independent = True: what a batch-1 call returns
    assert len(set(alone)) > 250, "the random model's transcripts should tell the signals apart"
    wrong, done, errors = [], [0] * n_threads, []
    stop = time.time() + seconds

    with BatchingTranscriber(launch_batch=eng.launch, max_batch=64, max_wait_ms=2.0, policy="independent", max_pad_ratio=1e9) as srv:
        def client(k):
            r = np.random.default_rng(1000 * seed + k)
            try:
                while time.time() < stop:
                    burst = [int(r.integers(0, len(pool))) for _ in range(int(r.integers(1, 9)))]
                    futs = [(i, srv.submit(pool[i])) for i in burst]
                    for i, f in futs:
                        t = f.result(120)
                        done[k] += 1
                        if t != alone[i]:
                            wrong.append((k, i, t[:30], alone[i][:30]))
                    if r.random() < 0.2:
                        time.sleep(float(r.random()) * 0.003)
            except Exception as e:  # noqa: BLE001
                errors.append(repr(e)[:200])
        ths = [threading.Thread(target=client, args=(k,)) for k in range(n_threads)]
        t0 = time.time()
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        el = time.time() - t0
        stats = dict(srv.stats)
    sizes = stats.get("device_calls_by_size", {})
    print(json.dumps({"seconds": round(el, 1), "threads": n_threads, "requests": int(sum(done)), "wrong": len(wrong), "errors": errors[:3],
                      "first_wrong": wrong[:3], "batches": stats.get("batches"), "largest_batch": max(sizes) if sizes else 0,
                      "requests_per_s": round(sum(done) / el, 1)}))


if __name__ == "__main__":
    main()
