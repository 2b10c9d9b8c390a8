#!/usr/bin/env python3
"""Concurrency stress below Python (dev tool, GPU): python tests/devtools/stress_threads.py [seconds] [threads] [seed]

include/vasr.h: "calls on distinct handles, or on one handle with distinct workspaces and streams, may run concurrently; there is no
global mutable state besides the (thread-local) error string".  Here `threads` host threads, each on a HIP stream of its own, call
vasr_transcribe_greedy_f32 / _pcm16 back to back (ctypes releases the GIL: the launches really interleave) -- half of them on
engines of their own (distinct handles), the other half on shallow copies of ONE engine (the same handle, its own workspace each)
-- over a pool of batches of 1-48 rows; every result (ids, id_len, log-probs) must be, bit for bit, what the same batch gave when
it ran alone before the threads started.  Prints one JSON line."""
import copy
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


def main():
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    n_threads = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    seed = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    cfg = configs.builtin("quartznet12x1_vi")
    jas = cfg["JasperEncoder"]["jasper"]
    enc_sd, dec_sd = synth.encoder_state_dict(jas, 64, 5), synth.decoder_state_dict(1024, len(cfg["labels"]) + 1, 5)
    shared = QuartzNetCTC(cfg, enc_sd, dec_sd)
    rng = np.random.default_rng(seed)
    pool = []
    for i in range(40):
        B = int(rng.choice([1, int(rng.integers(2, 9)), int(rng.integers(9, 49))]))
        L = int(rng.integers(4000, 60000)) if B < 9 else int(rng.integers(4000, 24000))
        sig, lens = synth.audio_batch(B, L, 1000 + i, ragged=True)
        if i % 4 == 0:
            sig = np.clip(np.round(sig * 32768 * 3), -32768, 32767).astype(np.int16)
        pool.append((torch.from_numpy(sig).cuda(), torch.from_numpy(lens).cuda()))
    want = []
    for w, n in pool:
        r = shared.forward(w, n, want_logp=True)
        torch.cuda.synchronize()
        want.append((r["ids"].clone(), r["id_len"].clone(), r["logp"].clone()))
    engines = []
    for k in range(n_threads):
        if k % 2 == 0:
            engines.append(QuartzNetCTC(cfg, enc_sd, dec_sd))              # a handle of its own
        else:
            e = copy.copy(shared)                                          # the SAME handle, a workspace of its own
            e._ws = None
            engines.append(e)
    wrong, done, errors = [], [0] * n_threads, []
    stop = time.time() + seconds

    def worker(k):
        r_ = np.random.default_rng(100 * seed + k)
        st = torch.cuda.Stream()
        try:
            with torch.cuda.stream(st):
                while time.time() < stop:
                    i = int(r_.integers(0, len(pool)))
                    w, n = pool[i]
                    r = engines[k].forward(w, n, want_logp=True)
                    st.synchronize()
                    ids, nn, lp = want[i]
                    ok = torch.equal(r["id_len"], nn) and torch.equal(r["logp"], lp) and all(
                        torch.equal(r["ids"][b, : int(nn[b])], ids[b, : int(nn[b])]) for b in range(ids.shape[0]))
                    done[k] += 1
                    if not ok:
                        d = (r["logp"] - lp).abs()
                        rows = torch.nonzero(d.amax((1, 2)) > 0).flatten().tolist()
                        wrong.append((k, i, int(w.shape[0]), "int16" if w.dtype == torch.int16 else "f32", f"{float(d.max()):.2e}", rows[:6]))
        except Exception as e:  # noqa: BLE001
            errors.append(f"thread {k}: {e!r}"[:300])

    ths = [threading.Thread(target=worker, args=(k,)) for k in range(n_threads)]
    t0 = time.time()
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    print(json.dumps({"seconds": round(time.time() - t0, 1), "threads": n_threads, "calls": int(sum(done)), "wrong": len(wrong),
                      "first_wrong": wrong[:5], "errors": errors[:3], "calls_per_thread": done}))


if __name__ == "__main__":
    main()
