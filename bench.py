#!/usr/bin/env python3
"""bench.py -- whole-job throughput of the viet-asr hot path on N MI355X of one node.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config {2,3,4,5}] [--gemm ...]

One *step* = one pass of the hot path over one batch of synthetic audio per GPU (input already resident in HBM),
then the gather of the collapsed id sequences to every rank (the reference's _infer gathers each returned tensor with
all_gather, nemo/backends/pytorch/actions.py:774-807).  Utterances are independent, so ranks shard them with no other
data-path collective: weak scaling, per-GPU work fixed.

Workloads (BASELINE.json `configs`, 0-based):
  --config 3 (default)  configs[2]  QuartzNet15x5 greedy CTC, 64 x 10 s per GPU -- the configuration the metric and the
                                    north-star target are quoted on
  --config 2            configs[1]  QuartzNet12x1 (Vietnamese head), 32 x 10 s, greedy
  --config 4            configs[3]  QuartzNet15x5 log-probs -> device beam search, beam_width 128, with a synthetic 3-gram
                                    ARPA model of ~1.2e5 n-grams (the reference's KenLM binaries are absent), 64 x 10 s;
                                    the search of batch k runs on a side stream under the acoustic pass of batch k + 1
  --config 5            configs[4]  one GPU's shard of the 8-GPU job: 512 clips of 30 s at 8 kHz -> device resampling to
                                    16 kHz -> QuartzNet15x5 greedy

`python bench.py --gpus N` with N > 1 starts its own N ranks (re-executes itself under torch.distributed.run on a free
port of 127.0.0.1); launched by torchrun / the driver it uses the ranks it is given.  One process per GPU, backend
"nccl" = RCCL over xGMI.

Prints ONE JSON line on rank 0 with the driver contract keys plus
  roofline      -- dominant kernel (1x1-conv GEMM): executed MFMA flops / summed kernel durations (per-launch dispatch
                   timestamps, hipExtLaunchKernelGGL) vs the dense MFMA peak of the operand type
  depthwise     -- depthwise-conv kernels: algorithmic HBM bytes / summed kernel durations vs 8 TB/s
  beam          -- (--config 4) the search kernel: ms per batch, workgroups (= CUs) it occupies
  cpu_baseline  -- the CPU oracle (same ATen ops as the reference) on this box's host cores, bounded sample
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import viet_asr_amd  # noqa: E402,F401
from viet_asr_amd import _lib, configs, synth  # noqa: E402
from viet_asr_amd.engine import QuartzNetCTC  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md "Peak FP32 (matrix)"
PEAK_16BIT_MFMA_TFLOPS = 2500.0  # same guide, "Peak BF16/FP16 MFMA" dense
PEAK_HBM_GBS = 8000.0          # same guide, HBM3E spec peak

WORKLOADS = {   # config -> (model, batch per GPU, clip seconds, input rate, decoder)
    2: ("quartznet12x1_vi", 32, 10.0, 16000, "greedy"),
    3: ("quartznet15x5", 64, 10.0, 16000, "greedy"),
    4: ("quartznet15x5", 64, 10.0, 16000, "beam"),
    5: ("quartznet15x5", 512, 30.0, 8000, "greedy"),
}
# MFMA products issued per fp32 multiply-add, operand type, dtype string of the JSON line
GEMM_MODES = {
    "f16x2": (3.0, "f16", "f32 via 2xf16 split operands (22-bit significands, 3 f16 MFMA products per multiply, fp32 accumulate)"),
    "bf16x3": (6.0, "bf16", "f32 via 3xbf16 split operands (6 bf16 MFMA products per multiply, fp32 accumulate)"),
    "bf16x2": (3.0, "bf16", "REDUCED: 2xbf16 split operands (16-bit significands, 3 bf16 MFMA products, fp32 accumulate) -- "
                            "opt-in mode, not the headline configuration"),
    "fp32": (1.0, "f32", "f32"),
}


def pmc_traffic(prefix, suffix=""):
    """Launch-weighted mean HBM bytes per launch of the kernels whose name starts with ``prefix``, from the newest
    COMMITTED PMC summary (separate --pmc FETCH_SIZE / WRITE_SIZE passes of this same command, tools/profile_round.sh;
    FETCH_SIZE doubled per MI355X_MICROARCH.md §HBM).  An offline figure: reported under traffic_offline, never as a
    live measurement.  None when no summary exists."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic_summary.json")))
    if not files:
        return None, None
    d = json.load(open(files[-1]))
    n = b = 0
    for k, v in d.items():
        if k.startswith(prefix) and k.endswith(suffix) and "fetch_bytes_corrected_per_launch" in v and "write_bytes_per_launch" in v:
            n += v["launches"]
            b += v["launches"] * (v["fetch_bytes_corrected_per_launch"] + v["write_bytes_per_launch"])
    return (round(b / n) if n else None), os.path.basename(files[-1])


def cpu_baseline(model, seed, decoder, lm_path, budget_s=25.0):
    """Time the CPU oracle (port of the reference path on the same ATen CPU ops) on a bounded sample of the workload:
    best of up to 5 runs after one warm-up, inside ~25 s; the intra-op thread count in use is reported."""
    from oracle import quartznet_oracle as O   # checker / baseline only -- never on the product path
    cfg = configs.builtin(model)
    jas = cfg["JasperEncoder"]["jasper"]
    enc_sd = synth.encoder_state_dict(jas, 64, seed)
    dec_sd = synth.decoder_state_dict(jas[-1]["filters"], len(cfg["labels"]) + 1, seed)
    # torch's own default intra-op thread count (it follows the process's CPU affinity / cgroup); forcing os.cpu_count()
    # oversubscribes a containerised box (measured: 256 threads on the GPU box -> 100x slower, 13 minutes for five runs)
    cores = torch.get_num_threads()
    if decoder == "beam":
        from oracle import beam_oracle as BO
        b, seconds = 1, 2.0       # the restated pyctcdecode loop is pure Python: one 2 s utterance is ~10 s of CPU
        lm = BO.LanguageModel(BO.NgramLM.from_arpa(lm_path), alpha=0.5, beta=1.5) if lm_path else None
    else:
        b, seconds = 4, 10.0
    sig, lens = synth.audio_batch(b, int(seconds * 16000), seed, ragged=False)

    def once():
        r = O.forward_all(sig, lens, enc_sd, dec_sd, jas)
        if decoder == "beam":
            return [BO.decode(r["logp"][i].numpy(), cfg["labels"], 128, lm=lm)
                    for i in range(b)]
        return O.ctc_decode_strings(r["pred"], cfg["labels"])

    with torch.no_grad():
        t0 = time.perf_counter()
        once()                                                   # warm-up
        warm = time.perf_counter() - t0
        times = [warm] if warm > budget_s / 2 else []    # a box this slow: the warm-up run is the sample
        t_all = time.perf_counter()
        # best of up to 5 runs, but never past the time budget (a slow box gets fewer runs, not a longer bench)
        while len(times) < 5 and (not times or (warm <= budget_s / 2 and warm + (time.perf_counter() - t_all) + min(times) < budget_s)):
            t0 = time.perf_counter()
            once()
            times.append(time.perf_counter() - t0)
    best = min(times)
    return {"value": round(b * seconds / best, 2), "unit": "audio-sec/wall-sec", "cores": cores, "kind": "port",
            "sample": f"{model} {decoder}{' beam 128 + 3-gram LM' if decoder == 'beam' else ''}, batch {b} x {seconds:g} s, "
                      f"best of {len(times)} after 1 warm-up, {cores} intra-op threads (torch default)"
                      + (" (acoustic model on all cores, the search loop is single-threaded Python)" if decoder == "beam" else ""),
            "utts_per_sec": round(b / best, 3), "median_value": round(b * seconds / float(np.median(times)), 2)}


def self_launch(n, argv):
    """`python bench.py --gpus N` without a launcher: start N ranks of this script under torch.distributed.run."""
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < n:
        sys.stderr.write(f"bench.py: --gpus {n} needs {n} visible HIP devices, found {have}; nothing was run\n")
        sys.exit(3)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, VASR_BENCH_LAUNCHED="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + [a for a in argv if a != "--spawn"]
    sys.exit(subprocess.run(cmd, env=env).returncode)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=int, choices=sorted(WORKLOADS), default=3, help="BASELINE.json workload (see docstring)")
    ap.add_argument("--model", default=None, help="override the workload's model")
    ap.add_argument("--batch", type=int, default=None, help="override the workload's batch per GPU")
    ap.add_argument("--seconds", type=float, default=None, help="override the workload's clip length")
    ap.add_argument("--ragged", action="store_true", help="lengths uniform in [L/2, L] instead of full clips")
    ap.add_argument("--beam-width", type=int, default=128)
    ap.add_argument("--no-lm", action="store_true", help="--config 4 without the n-gram model")
    ap.add_argument("--no-overlap", action="store_true", help="--config 4: search on the acoustic stream (serialised)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-gemm", action="store_true", help="skip the comparison pass in the other GEMM arithmetic")
    ap.add_argument("--spawn", action="store_true", help="go through the self-launcher even for --gpus 1 (RCCL world of 1)")
    ap.add_argument("--gemm", choices=sorted(GEMM_MODES), default=None,
                    help="arithmetic of the 1x1-conv GEMMs (default: the library's default mode)")
    a = ap.parse_args()

    launched = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if not launched and (a.gpus > 1 or a.spawn):
        self_launch(a.gpus, sys.argv[1:])
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        sys.exit(f"bench.py: --gpus {a.gpus} but the launcher started {world} ranks")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    if world > 1 or os.environ.get("VASR_BENCH_FORCE_DIST") or os.environ.get("VASR_BENCH_LAUNCHED"):
        import torch.distributed as dist_
        for k, v in (("RANK", "0"), ("WORLD_SIZE", "1"), ("MASTER_PORT", "29533")):
            os.environ.setdefault(k, v)
        dist = dist_
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", init_method="env://", device_id=dev)   # RCCL over xGMI

    model, batch, seconds, rate, decoder = WORKLOADS[a.config]
    model, batch, seconds = a.model or model, a.batch or batch, a.seconds or seconds
    seed = 3
    cfg = configs.builtin(model)
    jas = cfg["JasperEncoder"]["jasper"]
    enc_sd = synth.encoder_state_dict(jas, 64, seed)
    dec_sd = synth.decoder_state_dict(jas[-1]["filters"], len(cfg["labels"]) + 1, seed)
    eng = QuartzNetCTC(cfg, enc_sd, dec_sd, device=dev, gemm=a.gemm)
    gemm = a.gemm or eng.handle.gemm_mode_name()
    in_samples = int(seconds * rate)
    sig, lens = synth.audio_batch(batch, in_samples, seed + 100 * rank, ragged=a.ragged)
    wav_in = torch.from_numpy(sig).to(dev)
    ln_in = torch.from_numpy(lens).to(dev)
    audio_sec_per_step = float(lens.sum()) / rate
    samples = int(seconds * 16000)               # at the model's rate

    beam_dec, lm_path, lm_info = None, None, None
    if decoder == "beam":
        from viet_asr_amd.beam import BeamSearchDecoder
        if not a.no_lm:        # every rank writes its own copy (deterministic, < 1 s)
            lm_path = os.path.join(tempfile.mkdtemp(prefix="vasr_lm_"), "synthetic3.arpa")
            synth.synthetic_arpa(lm_path, cfg["labels"], seed=seed)
        beam_dec = BeamSearchDecoder(cfg["labels"], lm_path=lm_path, alpha=0.5, beta=1.5)
        lm = beam_dec._get_lm()
        if lm is not None:
            lm_info = {"order": lm.order, "ngrams": lm.n_ngrams, "words": lm.n_words, "table_load": round(lm.table_load, 3)}

    if rate != 16000:
        from viet_asr_amd import audio

    def acoustic_input():
        if rate == 16000:
            return wav_in, ln_in
        return audio.resample(wav_in, ln_in, rate, 16000)

    gathered, inflight, n_steps = None, [None, None], 0

    def drain(slot):
        if inflight[slot] is not None:
            for w in inflight[slot][:2]:
                w.wait()
            inflight[slot] = None

    def step():
        nonlocal gathered, n_steps
        wav, ln = acoustic_input()
        if decoder == "beam":
            r = eng.forward_beam(wav, ln, beam_dec, a.beam_width, overlap=not a.no_overlap)
        else:
            r = eng.forward(wav, ln, want_logp=False, want_pred=False)
        if dist is not None:
            # Result gather, one collective per returned tensor like actions.py:774-807.  Issued asynchronously on
            # RCCL's own stream into one of two buffers: the next batch's kernels do not wait for the other ranks, a
            # buffer is reused only after its previous gather has been waited for, and sync() drains both.
            if gathered is None:
                gathered = [(torch.empty((world,) + tuple(r["ids"].shape), dtype=torch.int32, device=dev),
                             torch.empty((world, batch), dtype=torch.int32, device=dev)) for _ in range(2)]
            slot = n_steps % 2
            n_steps += 1
            drain(slot)
            if r.get("done") is not None:
                torch.cuda.current_stream().wait_event(r["done"])     # the gather reads what the side stream wrote
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
    rccl = None
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
        ids_dev = torch.tensor([local], dtype=torch.int32, device=dev)
        all_dev = torch.empty((world,), dtype=torch.int32, device=dev)
        dist.all_gather_into_tensor(all_dev, ids_dev)
        rccl = {"rccl_ranks": dist.get_world_size(), "backend": dist.get_backend(), "rank_devices": all_dev.tolist()}
    else:
        audio_all = audio_sec_per_step

    # ---- second pass, same K steps, with per-kernel-class HIP events on the launch stream ----
    wav16, ln16 = acoustic_input()
    torch.cuda.synchronize()
    eng.handle.profile_begin()
    for _ in range(a.steps):
        eng.forward(wav16, ln16, want_logp=False, want_pred=False)
    torch.cuda.synchronize()
    prof = eng.handle.profile_end()
    # depthwise / pointwise launches carry their own (start, stop) events (hipExtLaunchKernelGGL: the dispatch packet's
    # begin / end timestamps), so these are kernel durations as rocprofv3 --kernel-trace reports them
    # (profiles/rNN_bench_kernel_stats.csv); the 2-3 us dispatch gap between dependent launches is in ms_per_step only.
    work = eng.handle.algorithmic_work(batch, samples)

    def timed(fn, n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        fn()
        torch.cuda.synchronize()
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n

    extra = {}
    if decoder == "beam" and rank == 0:
        lp = eng.forward(wav16, ln16, want_logp=True, want_pred=False)["logp"]
        ms_beam = timed(lambda: beam_dec.decode_ids(lp, a.beam_width), max(3, a.steps // 4))
        ms_ac = timed(lambda: eng.forward(wav16, ln16, want_logp=True, want_pred=False), max(3, a.steps // 4))
        n_cu = torch.cuda.get_device_properties(dev).multi_processor_count
        # the random-weight model's posteriors are near-deterministic (~1.1 classes per frame clear token_min_logp), which
        # is the cheapest input a beam search can get; the same search on CTC-like posteriors of the same shape that spell
        # words of the LM's vocabulary (beams branch, merge, LM re-ranks at word boundaries) is timed beside it
        ms_ctc = None
        if lm_path:
            from viet_asr_amd.beam import read_arpa
            words = sorted(w[0] for w in read_arpa(lm_path)[1] if len(w) == 1 and not w[0].startswith("<"))
            lp_ctc = torch.from_numpy(synth.ctc_like_log_probs(batch, lp.shape[1], cfg["labels"], words, seed=seed)).to(dev)
            ms_ctc = timed(lambda: beam_dec.decode_ids(lp_ctc, a.beam_width), max(3, a.steps // 4))
        extra["beam"] = {"kernel": "beam_search_kernel (one 512-thread workgroup per utterance, merge table in LDS)",
                         "bound": "LDS latency (dependent round trips per frame)", "beam_width": a.beam_width,
                         "ms_per_batch_alone": round(ms_beam, 3), "acoustic_ms_per_batch_alone": round(ms_ac, 3),
                         "ms_per_batch_on_ctc_like_posteriors": round(ms_ctc, 3) if ms_ctc else None,
                         "serial_sum_ms": round(ms_beam + ms_ac, 3), "workgroups": batch, "cus": n_cu,
                         "cus_busy_frac": round(min(1.0, batch / n_cu), 3), "overlapped": not a.no_overlap, "lm": lm_info}
    if rate != 16000 and rank == 0:
        ms_rs = timed(lambda: audio.resample(wav_in, ln_in, rate, 16000), max(3, a.steps // 4))
        in_b, out_b = wav_in.numel() * 4, wav16.numel() * 4
        extra["resample"] = {"kernel": "resample_poly_kernel (windowed-sinc, wings in LDS)", "ms_per_batch": round(ms_rs, 3),
                             "bound": "hbm", "achieved": round((in_b + out_b) / (ms_rs * 1e-3) / 1e9, 1), "peak": PEAK_HBM_GBS,
                             "unit": "GB/s", "frac": round((in_b + out_b) / (ms_rs * 1e-3) / 1e9 / PEAK_HBM_GBS, 4)}

    # ---- the other GEMM arithmetic on the same workload, for reference (rank 0 of a single-GPU run only) ----
    other = None
    if world == 1 and not a.no_other_gemm and decoder == "greedy" and rate == 16000:
        other_mode = "fp32" if gemm != "fp32" else "bf16x3"
        eng.handle.set_gemm_mode(other_mode)
        for _ in range(2):
            eng.forward(wav16, ln16, want_logp=False, want_pred=False)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(a.steps):
            eng.forward(wav16, ln16, want_logp=False, want_pred=False)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t1
        eng.handle.profile_begin()
        for _ in range(a.steps):
            eng.forward(wav16, ln16, want_logp=False, want_pred=False)
        torch.cuda.synchronize()
        p2 = eng.handle.profile_end()
        tf = work["pointwise_flops"] / (p2["pointwise"]["ms"] / a.steps * 1e-3) / 1e12
        other = {"gemm": other_mode, "value": round(audio_sec_per_step * a.steps / dt, 1),
                 "ms_per_step": round(dt / a.steps * 1e3, 3), "pointwise_ms_per_step": round(p2["pointwise"]["ms"] / a.steps, 3),
                 "pointwise_fp32_equivalent_tflops": round(tf, 2)}
        eng.handle.set_gemm_mode(gemm)

    if rank == 0:
        if r.get("done") is not None:
            r["done"].synchronize()
        hyp = eng.texts(r["ids"], r["id_len"])
        # The GEMM family = the 1x1-conv GEMM launches + the fused depthwise -> pointwise launches (encoder_fused.hip), whose
        # whole duration -- depthwise producers included -- is charged to the GEMM: work that ran / time it took, per class
        # (vasr_profile_end sums the algorithmic flops / bytes of the launches it bracketed)
        fu_ms = prof["fused"]["ms"] / a.steps
        pw_ms = prof["pointwise"]["ms"] / a.steps + fu_ms
        dw_ms = prof["depthwise"]["ms"] / a.steps
        gemm_flops = (prof["pointwise"]["flops"] + prof["fused"]["flops"]) / a.steps
        assert abs(gemm_flops - work["pointwise_flops"]) <= 1e-6 * work["pointwise_flops"], (gemm_flops, work["pointwise_flops"])
        pw_tflops = gemm_flops / (pw_ms * 1e-3) / 1e12       # fp32-equivalent (algorithmic) rate
        terms, optype, dtype_str = GEMM_MODES[gemm]
        # a split kernel executes `terms` 16-bit MFMA products per fp32 multiply-add: that is the work the matrix pipe sees
        exec_tflops = pw_tflops * terms
        peak = PEAK_F32_MFMA_TFLOPS if gemm == "fp32" else PEAK_16BIT_MFMA_TFLOPS
        dw_bytes = prof["depthwise"]["bytes"] / a.steps      # of the depthwise layers that ran as kernels of their own
        dw_gbs = dw_bytes / (dw_ms * 1e-3) / 1e9
        headline = a.config == 3 and not (a.model or a.batch or a.seconds or a.ragged)
        kname = "pw_gemm_kernel" if gemm == "fp32" else "pw_gemm_split_kernel"
        arith_tag = {"f16x2": ", 2>", "bf16x3": ", 0>", "bf16x2": ", 1>"}.get(gemm, "")
        pw_traffic, traffic_src = pmc_traffic(kname, arith_tag) if headline else (None, None)
        dw_traffic, _ = pmc_traffic("dw_") if headline else (None, None)
        what = {"greedy": "greedy CTC", "beam": f"beam search (width {a.beam_width}" + (", 3-gram LM" if lm_info else ", no LM") + ")"}[decoder]
        out = {
            "metric": "real_time_factor", "value": round(audio_all * a.steps / elapsed, 1),
            "unit": "audio-sec/wall-sec", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(elapsed / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": dtype_str, "data": "synthetic",
            "config": {"workload": f"BASELINE configs[{a.config - 1}]: {model} {what}, batch={batch}x{seconds:g}s "
                                   f"{rate // 1000}kHz mono per GPU{' (ragged lengths)' if a.ragged else ''}"
                                   f"{' -> 16 kHz on the device' if rate != 16000 else ''}, wav in HBM -> "
                                   f"{'collapsed ids' if decoder == 'greedy' else 'best-hypothesis ids'}",
                       "batch_per_gpu": batch, "clip_seconds": seconds, "parallelism": f"utterance-shard x{world}"},
            "utts_per_sec": round(batch * world * a.steps / elapsed, 1),
            "roofline": {"kernel": f"{kname} (1x1 conv as {'split-operand ' if terms > 1 else ''}MFMA GEMM + BN/residual/ReLU epilogue)",
                         "bound": "mfma", "achieved": round(exec_tflops, 2), "peak": peak,
                         "unit": "TFLOP/s", "frac": round(exec_tflops / peak, 4),
                         "mfma_products_per_multiply": terms, "fp32_equivalent_tflops": round(pw_tflops, 2),
                         # the same algorithmic work against the matrix pipe's fp32 peak (what an fp32-MFMA kernel could
                         # reach at most): > 1 means the split arithmetic beats the best possible exact-fp32 GEMM
                         "frac_of_fp32_mfma_peak": round(pw_tflops / PEAK_F32_MFMA_TFLOPS, 3),
                         "traffic": None, "traffic_offline": pw_traffic,
                         "traffic_note": "HBM bytes per launch from the committed PMC pass named in traffic_source (not measured in "
                                         "this run; null off the headline workload)", "traffic_source": traffic_src,
                         "flops_per_step": work["pointwise_flops"], "ms_per_step": round(pw_ms, 3),
                         "launches_per_step": (prof["pointwise"]["launches"] + prof["fused"]["launches"]) // a.steps},
            "fused": {"kernel": "dwpw_fused_kernel<K, DUAL> (depthwise + 1x1 conv + BN + residual + ReLU of a 256-channel sub-block in one "
                                "launch; included in roofline above)", "launches_per_step": prof["fused"]["launches"] // a.steps,
                      "ms_per_step": round(fu_ms, 3), "flops_per_step": prof["fused"]["flops"] / a.steps,
                      "hbm_bytes_per_step": prof["fused"]["bytes"] / a.steps,
                      "achieved_GBps": round(prof["fused"]["bytes"] / a.steps / (fu_ms * 1e-3) / 1e9, 1) if fu_ms else None,
                      "achieved_TFLOPs_executed": round(prof["fused"]["flops"] / a.steps * terms / (fu_ms * 1e-3) / 1e12, 1) if fu_ms else None},
            "depthwise": {"kernel": "depthwise conv kernels (dw_toeplitz_kernel<K,DIL> on the matrix pipe; dw_conv_generic for the stride-2 prologue; dw_pair_kernel under --gemm fp32 | bf16x3 or VASR_DW_MFMA=0)", "bound": "hbm", "achieved": round(dw_gbs, 1),
                          "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(dw_gbs / PEAK_HBM_GBS, 4), "traffic": None,
                          "traffic_offline": dw_traffic, "bytes_per_step": dw_bytes, "ms_per_step": round(dw_ms, 3),
                          "launches_per_step": prof["depthwise"]["launches"] // a.steps},
            "other_ms_per_step": {"frontend": round(prof["frontend"]["ms"] / a.steps, 3),
                                  "head": round(prof["head"]["ms"] / a.steps, 3)},
            "sample_transcript": hyp[0][:32],
        }
        out.update(extra)
        if rccl is not None:
            out.update(rccl)
        if other is not None:
            out["other_gemm_arithmetic"] = other
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(model, seed, decoder, lm_path)
        if dist is not None:
            import ctypes
            ctypes.CDLL(None).fflush(None)      # RCCL's version banner sits in C stdio's buffer: keep the JSON line last
        print(json.dumps(out, ensure_ascii=False), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
