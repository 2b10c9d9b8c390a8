#!/usr/bin/env python3
"""bench.py -- whole-job throughput of the viet-asr hot path on N MI355X of one node.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--model quartznet15x5] [--batch 64] [--seconds 10]

One *step* = one pass of the hot path (wav already resident in HBM -> mel -> QuartzNet encoder ->
CTC head -> greedy argmax -> CTC collapse) over one batch of synthetic 16 kHz audio per GPU, then
the gather of the collapsed id sequences to every rank (the reference's _infer gathers each
returned tensor with all_gather, nemo/backends/pytorch/actions.py:774-807).  Utterances are
independent, so ranks shard them with no other data-path collective: weak scaling, per-GPU work
fixed.  Workload at any N: BASELINE.json configs[2] = QuartzNet15x5, batch 64 x 10 s per GPU
(the configuration the metric/target is quoted on).

Prints ONE JSON line on rank 0 with the driver contract keys plus
  roofline      -- dominant kernel (1x1-conv GEMM, fp32 operands as 3 x bf16 on the bf16 MFMA pipe): executed flops /
                   summed kernel durations (per-launch dispatch timestamps, hipExtLaunchKernelGGL) vs the 2.5 PFLOP/s
                   nominal peak, plus the rate a bare MFMA stream sustains on this box (sustained_peak)
  depthwise     -- depthwise-conv kernels: algorithmic HBM bytes / summed kernel durations vs 8 TB/s
  cpu_baseline  -- the CPU oracle (same ATen ops as the reference) on this box's host cores
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import viet_asr_amd  # noqa: E402,F401
from viet_asr_amd import _lib, configs, synth  # noqa: E402
from viet_asr_amd.engine import QuartzNetCTC  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md "Peak FP32 (matrix)"
PEAK_BF16_MFMA_TFLOPS = 2500.0 # same guide, "Peak BF16/FP16 MFMA" dense
PEAK_HBM_GBS = 8000.0          # same guide, HBM3E spec peak


def pmc_traffic(prefix):
    """Launch-weighted mean HBM bytes per launch of the kernels whose name starts with ``prefix``, from the newest
    committed PMC summary (separate --pmc FETCH_SIZE / WRITE_SIZE passes of this same command,
    tools/profile_round.sh; FETCH_SIZE doubled per MI355X_MICROARCH.md §HBM).  None when no summary exists."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic_summary.json")))
    if not files:
        return None, None
    d = json.load(open(files[-1]))
    n = b = 0
    for k, v in d.items():
        if k.startswith(prefix) and "fetch_bytes_corrected_per_launch" in v and "write_bytes_per_launch" in v:
            n += v["launches"]
            b += v["launches"] * (v["fetch_bytes_corrected_per_launch"] + v["write_bytes_per_launch"])
    return (round(b / n) if n else None), os.path.basename(files[-1])


def cpu_baseline(model, seed, seconds, budget_s=20.0):
    """Time the CPU oracle (port of the reference path on the same ATen CPU ops) on a bounded sample."""
    from oracle import quartznet_oracle as O   # checker / baseline only -- never on the product path
    cfg = configs.builtin(model)
    jas = cfg["JasperEncoder"]["jasper"]
    enc_sd = synth.encoder_state_dict(jas, 64, seed)
    dec_sd = synth.decoder_state_dict(jas[-1]["filters"], len(cfg["labels"]) + 1, seed)
    b = 8
    sig, lens = synth.audio_batch(b, int(seconds * 16000), seed, ragged=False)
    cores = torch.get_num_threads()
    with torch.no_grad():
        t0 = time.perf_counter()
        O.forward_all(sig, lens, enc_sd, dec_sd, jas)          # warm-up
        warm = time.perf_counter() - t0
        times = []
        while len(times) < 3 and sum(times) + warm < budget_s:
            t0 = time.perf_counter()
            r = O.forward_all(sig, lens, enc_sd, dec_sd, jas)
            O.ctc_decode_strings(r["pred"], cfg["labels"])
            times.append(time.perf_counter() - t0)
    best = min(times) if times else warm
    return {"value": round(b * seconds / best, 2), "unit": "audio-sec/wall-sec", "cores": cores, "kind": "port",
            "sample": f"{model} greedy, batch {b} x {seconds:g} s, best of {max(len(times), 1)} after 1 warm-up",
            "utts_per_sec": round(b / best, 3)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--model", default="quartznet15x5")
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--seconds", type=float, default=10.0)
    ap.add_argument("--ragged", action="store_true", help="lengths uniform in [L/2, L] instead of full clips")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-gemm", action="store_true", help="skip the comparison pass in the other GEMM arithmetic")
    ap.add_argument("--gemm", choices=["bf16x3", "fp32", "bf16x2"], default="bf16x3",
                    help="arithmetic of the 1x1-conv GEMMs: 3 x bf16 split operands on the bf16 MFMA pipe "
                         "(fp32-equivalent accuracy, default) or exact-fp32 MFMA")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            sys.exit("launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node N "
                     "--master-addr 127.0.0.1 --master-port P bench.py --gpus N ...")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1 or os.environ.get("VASR_BENCH_FORCE_DIST"):      # the switch runs the gather path on a 1-GPU box
        import torch.distributed as dist_
        for k, v in (("RANK", "0"), ("WORLD_SIZE", "1"), ("MASTER_PORT", "29533")):
            os.environ.setdefault(k, v)
        dist = dist_
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", init_method="env://", device_id=dev)   # RCCL over xGMI

    seed = 3
    cfg = configs.builtin(a.model)
    jas = cfg["JasperEncoder"]["jasper"]
    enc_sd = synth.encoder_state_dict(jas, 64, seed)
    dec_sd = synth.decoder_state_dict(jas[-1]["filters"], len(cfg["labels"]) + 1, seed)
    eng = QuartzNetCTC(cfg, enc_sd, dec_sd, device=dev, gemm=a.gemm)
    samples = int(a.seconds * 16000)
    sig, lens = synth.audio_batch(a.batch, samples, seed + 100 * rank, ragged=a.ragged)
    wav = torch.from_numpy(sig).to(dev)
    ln = torch.from_numpy(lens).to(dev)
    audio_sec_per_step = float(lens.sum()) / 16000.0

    gathered, inflight, n_steps = None, [None, None], 0

    def drain(slot):
        if inflight[slot] is not None:
            for w in inflight[slot][:2]:
                w.wait()
            inflight[slot] = None

    def step():
        nonlocal gathered, n_steps
        r = eng.forward(wav, ln, want_logp=False, want_pred=False)
        if dist is not None:
            # Result gather, one collective per returned tensor like actions.py:774-807.  Issued asynchronously on
            # RCCL's own stream into one of two buffers: the next batch's kernels do not wait for the other ranks, a
            # buffer is reused only after its previous gather has been waited for, and sync() drains both.
            if gathered is None:
                gathered = [(torch.empty((world,) + tuple(r["ids"].shape), dtype=torch.int32, device=dev),
                             torch.empty((world, a.batch), dtype=torch.int32, device=dev)) for _ in range(2)]
            slot = n_steps % 2
            n_steps += 1
            drain(slot)
            inflight[slot] = (dist.all_gather_into_tensor(gathered[slot][0], r["ids"], async_op=True),
                              dist.all_gather_into_tensor(gathered[slot][1], r["id_len"], async_op=True), r)
        return r

    def sync():
        if dist is not None:
            drain(0)
            drain(1)
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        r = step()
    sync()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        last = (n_steps - 1) % 2       # the gathered copy of this rank's last batch must be what the engine returned
        if not (torch.equal(gathered[last][0][rank], r["ids"]) and torch.equal(gathered[last][1][rank], r["id_len"])):
            raise RuntimeError("result gather returned something else than this rank's own shard at its index")
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        tot = torch.tensor([audio_sec_per_step], dtype=torch.float64, device=dev)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        audio_all = float(tot.item())
    else:
        audio_all = audio_sec_per_step

    # ---- second pass, same K steps, with per-kernel-class HIP events on the launch stream ----
    eng.handle.profile_begin()
    for _ in range(a.steps):
        eng.forward(wav, ln, want_logp=False, want_pred=False)
    torch.cuda.synchronize()
    prof = eng.handle.profile_end()
    # depthwise / pointwise launches carry their own (start, stop) events (hipExtLaunchKernelGGL: the dispatch packet's
    # begin / end timestamps), so these are kernel durations as rocprofv3 --kernel-trace reports them
    # (profiles/rNN_bench_kernel_stats.csv); the 2-3 us dispatch gap between dependent launches is in ms_per_step only.
    work = eng.handle.algorithmic_work(a.batch, samples)

    # ---- what the matrix pipe sustains on this box: the GEMM's MFMA stream alone (no loads, LDS, barriers) ----
    sustained = None
    if rank == 0 and a.gemm == "bf16x3":
        import ctypes
        sink = torch.zeros(16, device=dev)
        fl = ctypes.c_double()
        n_cu = torch.cuda.get_device_properties(dev).multi_processor_count
        st = torch.cuda.current_stream().cuda_stream
        run = lambda: _lib.check(_lib.lib().vasr_bench_mfma_bf16_sustained(n_cu, 2000, sink.data_ptr(), ctypes.byref(fl), st))
        run(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            run()
        e1.record(); torch.cuda.synchronize()
        sustained = 3 * fl.value / (e0.elapsed_time(e1) * 1e-3) / 1e12

    # ---- the other GEMM arithmetic on the same workload, for reference (rank 0 of a single-GPU run only) ----
    other = None
    if world == 1 and not a.no_other_gemm:
        other_mode = "fp32" if a.gemm == "bf16x3" else "bf16x3"
        eng.handle.set_gemm_mode(other_mode)
        for _ in range(2):
            eng.forward(wav, ln, want_logp=False, want_pred=False)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(a.steps):
            eng.forward(wav, ln, want_logp=False, want_pred=False)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t1
        eng.handle.profile_begin()
        for _ in range(a.steps):
            eng.forward(wav, ln, want_logp=False, want_pred=False)
        torch.cuda.synchronize()
        p2 = eng.handle.profile_end()
        tf = work["pointwise_flops"] / (p2["pointwise"]["ms"] / a.steps * 1e-3) / 1e12
        other = {"gemm": other_mode, "value": round(audio_sec_per_step * a.steps / dt, 1),
                 "ms_per_step": round(dt / a.steps * 1e3, 3), "pointwise_ms_per_step": round(p2["pointwise"]["ms"] / a.steps, 3),
                 "pointwise_fp32_equivalent_tflops": round(tf, 2)}
        eng.handle.set_gemm_mode(a.gemm)

    if rank == 0:
        hyp = eng.texts(r["ids"], r["id_len"])
        pw_ms = prof["pointwise"]["ms"] / a.steps
        dw_ms = prof["depthwise"]["ms"] / a.steps
        pw_tflops = work["pointwise_flops"] / (pw_ms * 1e-3) / 1e12       # fp32-equivalent (algorithmic) rate
        split = a.gemm in ("bf16x3", "bf16x2")
        terms = {"bf16x3": 6.0, "bf16x2": 3.0}.get(a.gemm, 1.0)
        # the split kernel executes 6 bf16 MFMA products per fp32 multiply-add: that is the work the matrix pipe sees
        exec_tflops = pw_tflops * terms
        peak = PEAK_BF16_MFMA_TFLOPS if split else PEAK_F32_MFMA_TFLOPS
        dw_gbs = work["depthwise_bytes"] / (dw_ms * 1e-3) / 1e9
        default_workload = a.model == "quartznet15x5" and a.batch == 64 and a.seconds == 10.0 and not a.ragged
        pw_traffic, traffic_src = pmc_traffic("pw_gemm_bf16x3" if split else "pw_gemm_kernel") if default_workload else (None, None)
        dw_traffic, _ = pmc_traffic("dw_pair_kernel") if default_workload else (None, None)
        out = {
            "metric": "real_time_factor", "value": round(audio_all * a.steps / elapsed, 1),
            "unit": "audio-sec/wall-sec", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(elapsed / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None,
            "dtype": {"bf16x3": "f32 via 3xbf16 split operands (6 bf16 MFMA products per multiply, fp32 accumulate)",
                      "bf16x2": "REDUCED: 2xbf16 split operands (16-bit significands, 3 bf16 MFMA products, fp32 accumulate) -- "
                                "opt-in mode, not the headline configuration", "fp32": "f32"}[a.gemm],
            "data": "synthetic",
            "config": {"workload": f"{a.model} greedy CTC, batch={a.batch}x{a.seconds:g}s 16kHz mono per GPU"
                                   f"{' (ragged lengths)' if a.ragged else ''}, wav in HBM -> collapsed ids",
                       "batch_per_gpu": a.batch, "clip_seconds": a.seconds, "parallelism": f"utterance-shard x{world}"},
            "utts_per_sec": round(a.batch * world * a.steps / elapsed, 1),
            "roofline": {"kernel": ("pw_gemm_bf16x3_kernel (1x1 conv as 3xbf16-split MFMA GEMM" if split else
                                    "pw_gemm_kernel (1x1 conv fp32 MFMA GEMM") + " + BN/residual/ReLU epilogue)",
                         "bound": "mfma", "achieved": round(exec_tflops, 2), "peak": peak,
                         "unit": "TFLOP/s", "frac": round(exec_tflops / peak, 4),
                         "fp32_equivalent_tflops": round(pw_tflops, 2), "traffic": pw_traffic,
                         "sustained_peak": round(sustained, 1) if sustained else None,
                         "frac_of_sustained_peak": round(exec_tflops / sustained, 4) if sustained else None,
                         "sustained_peak_note": "the same MFMA stream with no loads / LDS / barriers, measured in this run: "
                                                "what the chip holds under its power limit (nominal peak assumes 2.4 GHz)",
                         "traffic_unit": "HBM bytes per launch (PMC, offline pass)", "traffic_source": traffic_src,
                         "flops_per_step": work["pointwise_flops"], "ms_per_step": round(pw_ms, 3),
                         "launches_per_step": prof["pointwise"]["launches"] // a.steps},
            "depthwise": {"kernel": "dw_pair_kernel<K, DIL> (utterance-pair packed-FMA depthwise)", "bound": "hbm", "achieved": round(dw_gbs, 1),
                          "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(dw_gbs / PEAK_HBM_GBS, 4), "traffic": dw_traffic,
                          "bytes_per_step": work["depthwise_bytes"], "ms_per_step": round(dw_ms, 3),
                          "launches_per_step": prof["depthwise"]["launches"] // a.steps},
            "other_ms_per_step": {"frontend": round(prof["frontend"]["ms"] / a.steps, 3),
                                  "head": round(prof["head"]["ms"] / a.steps, 3)},
            "sample_transcript": hyp[0][:32],
        }
        if other is not None:
            out["other_gemm_arithmetic"] = other
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a.model, seed, a.seconds)
        if dist is not None:
            import ctypes
            ctypes.CDLL(None).fflush(None)      # RCCL's version banner sits in C stdio's buffer: keep the JSON line last
        print(json.dumps(out, ensure_ascii=False), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
