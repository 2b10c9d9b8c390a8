#!/usr/bin/env python3
"""Host-inclusive throughput of the greedy path (dev tool): numpy signals in host memory -> transcripts (str).

bench.py times the device path with the batch resident in HBM (the contract of `value`).  A server also pays for the
collate, PCIe and ids -> str; this script measures what is left of the device rate once those are included:
  old       what transcribe() did before the pipelined path: numpy collate -> pageable .to(device) -> forward -> .cpu()
  blocking  eng.launch(batch).texts() one batch at a time (pinned staging, nothing overlapped)
  pipelined eng.launch() two batches in flight (collate + H2D of batch k+1 under the kernels of batch k), fp32 and int16
  served    BatchingTranscriber(launch_batch=...) fed by client threads, one request per utterance
Usage: python tools/bench_serving.py [batch=64] [seconds=10] [batches=40]
"""
import os
import sys
import threading
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import viet_asr_amd  # noqa: E402,F401
from viet_asr_amd import configs, synth  # noqa: E402
from viet_asr_amd.engine import QuartzNetCTC  # noqa: E402
from viet_asr_amd.serving import BatchingTranscriber  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
SEC = float(sys.argv[2]) if len(sys.argv) > 2 else 10.0
NB = int(sys.argv[3]) if len(sys.argv) > 3 else 40

cfg = configs.builtin("quartznet15x5")
jas = cfg["JasperEncoder"]["jasper"]
eng = QuartzNetCTC(cfg, synth.encoder_state_dict(jas, 64, 3), synth.decoder_state_dict(1024, 29, 3))
sig, _ = synth.audio_batch(B, int(SEC * 16000), 5)
f32 = [np.ascontiguousarray(sig[i]) for i in range(B)]
i16 = [np.clip(np.round(s * 32768.0), -32768, 32767).astype(np.int16) for s in f32]


def report(name, seconds, n_utts, extra=""):
    print(f"{name:28s} {seconds / (n_utts / B) * 1e3:7.2f} ms per batch of {B}   {n_utts / seconds:9,.0f} utt/s   "
          f"{n_utts * SEC / seconds:10,.0f}x real time {extra}", flush=True)


def old_path(signals):
    lens = np.array([len(s) for s in signals], dtype=np.int64)
    batch = np.zeros((len(signals), int(lens.max())), dtype=np.float32)
    for i, s in enumerate(signals):
        batch[i, : len(s)] = s
    r = eng.forward(torch.from_numpy(batch).to(eng.device), torch.from_numpy(lens).to(eng.device))
    return eng.texts(r["ids"], r["id_len"])


# device-only reference point (inputs resident, no host work): what bench.py reports
wav = torch.from_numpy(sig).cuda()
ln = torch.full((B,), sig.shape[1], dtype=torch.int64, device="cuda")
for _ in range(3):
    eng.forward(wav, ln)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(NB):
    eng.forward(wav, ln)
torch.cuda.synchronize()
report("device only (bench.py)", time.perf_counter() - t, NB * B)

for name, fn, data in (("old: pageable, blocking", old_path, f32),
                       ("blocking, pinned fp32", lambda s: eng.launch(s).texts(), f32),
                       ("blocking, pinned int16", lambda s: eng.launch(s).texts(), i16)):
    fn(data), fn(data)
    t = time.perf_counter()
    for _ in range(NB):
        fn(data)
    report(name, time.perf_counter() - t, NB * B)

for name, data in (("pipelined fp32", f32), ("pipelined int16", i16)):
    eng.launch(data).texts()
    t = time.perf_counter()
    prev = None
    for _ in range(NB):
        cur = eng.launch(data)
        if prev is not None:
            prev.texts()
        prev = cur
    prev.texts()
    report(name, time.perf_counter() - t, NB * B)

# through the batcher: CLIENTS threads, each submits its share of NB*B requests and waits for every answer
CLIENTS = 8
for name, data in (("served fp32, 8 clients", f32), ("served int16, 8 clients", i16)):
    lat = []
    with BatchingTranscriber(launch_batch=eng.launch, max_batch=B, max_wait_ms=2.0) as srv:
        srv.transcribe(data[0])

        def client(k):
            mine = []
            per = NB * B // CLIENTS
            window = []
            for j in range(per):
                window.append((time.perf_counter(), srv.submit(data[(k + j) % B])))
                if len(window) >= 2 * B // CLIENTS * 2:            # keep about two batches per client outstanding
                    t0, f = window.pop(0)
                    f.result(60)
                    mine.append(time.perf_counter() - t0)
            for t0, f in window:
                f.result(60)
                mine.append(time.perf_counter() - t0)
            lat.extend(mine)

        t = time.perf_counter()
        th = [threading.Thread(target=client, args=(k,)) for k in range(CLIENTS)]
        [x.start() for x in th]
        [x.join() for x in th]
        dt = time.perf_counter() - t
        sizes = dict(sorted(srv.stats["device_calls_by_size"].items()))
    lat = np.sort(np.array(lat)) * 1e3
    report(name, dt, len(lat), f"latency p50 {lat[len(lat) // 2]:.1f} ms p99 {lat[int(len(lat) * 0.99)]:.1f} ms; "
                               f"device calls by size {sizes}")

# mixed lengths (2-10 s, what a service sees): "exact" can only merge equal lengths, i.e. runs batch 1; "independent"
# merges everything and still returns the batch-1 answers (vasr_set_row_independent)
rng = np.random.default_rng(11)
mixed = [np.ascontiguousarray(f32[i % B][: int(rng.integers(32000, 160001))]) for i in range(4 * B)]
audio_s = sum(len(s) for s in mixed) / 16000.0
for policy, n_req in (("exact", 256), ("padded", len(mixed) * 4), ("independent", len(mixed) * 4)):
    with BatchingTranscriber(launch_batch=eng.launch, max_batch=B, max_wait_ms=2.0, policy=policy, max_pad_ratio=1.25) as srv:
        srv.transcribe(mixed[0])
        t = time.perf_counter()
        futs = [srv.submit(mixed[i % len(mixed)]) for i in range(n_req)]
        [f.result(120) for f in futs]
        dt = time.perf_counter() - t
        sizes = srv.stats["device_calls_by_size"]
    secs = sum(len(mixed[i % len(mixed)]) for i in range(n_req)) / 16000.0
    print(f"mixed 2-10 s, policy {policy:12s} {n_req / dt:8,.0f} utt/s  {secs / dt:9,.0f}x real time   "
          f"mean device batch {sum(k * v for k, v in sizes.items()) / sum(sizes.values()):.1f}", flush=True)
