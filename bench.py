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
  --config 5            configs[4]  the 8-GPU job: 4096 clips of 30 s at 8 kHz -> device resampling to 16 kHz ->
                                    QuartzNet15x5 greedy.  With --gpus 1 a step is ONE GPU's shard (512 clips, the weak-
                                    scaling form of every other workload); with --gpus N > 1 a step is the WHOLE 4096-clip
                                    job, sharded over the ranks (4096 / N clips each, in passes of <= 512), "scaling":
                                    "strong", and the line carries the ranks' min / max milliseconds.  --ragged draws clip
                                    lengths in [15 s, 30 s] and deals them to the ranks with dist.balanced_shards.

The default invocation (config 3, N = 1) also runs 5 steps each of configs 2, 4 and 5 after the headline and adds them
as "configs": {"2": {...}, "3r": {...}, "4": {...}, "5": {...}} to the SAME JSON line ("3r" = the headline shape with ragged
lengths U[L/2, L]), a "latency" block (batch 1 / 8 of 15x5, and the reference's own serving shape: batch 1, 12x1_vi, beam
50 / 100 + LM, split into acoustic pass and search) and a "sol" block (speed-of-light floors of the headline step).

`python bench.py --gpus N` with N > 1 starts its own N ranks (re-executes itself under torch.distributed.run on a free
port of 127.0.0.1); launched by torchrun / the driver it uses the ranks it is given.  One process per GPU, backend
"nccl" = RCCL over xGMI.

Prints ONE JSON line on rank 0 with the driver contract keys plus
  roofline      -- dominant kernel family (1x1-conv GEMMs, incl. the fused depthwise + pointwise launches): executed MFMA
                   flops of the launches that ran / their summed kernel durations (per-launch dispatch timestamps,
                   hipExtLaunchKernelGGL) vs the dense MFMA peak of the operand type
  depthwise     -- depthwise-conv kernels: algorithmic HBM bytes / summed kernel durations vs 8 TB/s (and vs the guide's
                   achievable ~6.3 TB/s: frac_of_achievable)
  fused         -- the fused depthwise + pointwise launches on their own
  beam          -- (--config 4) the search kernel: ms per batch, workgroups (= CUs) it occupies
  cpu_baseline  -- the CPU oracle (same ATen ops as the reference) on this box's host cores, bounded sample
  box           -- what THIS box sustains, measured after the timed region (bare MFMA stream, float4 streaming pass beyond and
                   inside the Infinity Cache); roofline.frac_of_measured / depthwise.frac_of_measured_copy* and the sol floors
                   are quoted against them, so that lines from different boxes can be compared
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
ACHIEVABLE_HBM_GBS = 6300.0    # same guide: what a streaming kernel sustains once its tensors outgrow the Infinity Cache

WORKLOADS = {   # config -> (model, batch per GPU, clip seconds, input rate, decoder)
    2: ("quartznet12x1_vi", 32, 10.0, 16000, "greedy"),
    3: ("quartznet15x5", 64, 10.0, 16000, "greedy"),
    4: ("quartznet15x5", 64, 10.0, 16000, "beam"),
    5: ("quartznet15x5", 512, 30.0, 8000, "greedy"),
}
JOB_CLIPS = int(os.environ.get("VASR_BENCH_JOB_CLIPS", "4096"))     # configs[4]: the whole 8-GPU job (env: tests shrink it)
# MFMA products issued per fp32 multiply-add, operand type, dtype string of the JSON line
GEMM_MODES = {
    "f16x2": (3.0, "f16", "f32 via 2xf16 split operands (22-bit significands, 3 f16 MFMA products per multiply, fp32 accumulate)"),
    "bf16x3": (6.0, "bf16", "f32 via 3xbf16 split operands (6 bf16 MFMA products per multiply, fp32 accumulate)"),
    "bf16x2": (3.0, "bf16", "REDUCED: 2xbf16 split operands (16-bit significands, 3 bf16 MFMA products, fp32 accumulate) -- "
                            "opt-in mode, not the headline configuration"),
    "fp32": (1.0, "f32", "f32"),
}


def pmc_traffic(prefix, suffix="", workload=""):
    """Launch-weighted mean HBM bytes per launch of the kernels whose name starts with ``prefix``, from the newest
    COMMITTED PMC summary (separate --pmc FETCH_SIZE / WRITE_SIZE passes of this same command, `bash tools/gpu.sh prof`;
    FETCH_SIZE doubled per MI355X_MICROARCH.md §HBM).  An offline figure: reported under traffic_offline, never as a
    live measurement.  None when no summary exists.  workload: "" = the headline (profiles/rNN_pmc_traffic_summary.json), "c5_" =
    one GPU's 512 x 30 s shard of configs[4] (profiles/rNN_c5_pmc_traffic_summary.json: tensors 12 x the Infinity Cache, so
    FETCH_SIZE there is HBM traffic, where the headline's counts hits in the 256 MiB cache as well)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r[0-9][0-9]_{workload}pmc_traffic_summary.json")))
    if not files:
        return None, None
    d = json.load(open(files[-1]))
    n = b = 0
    for k, v in d.items():
        if k.startswith(prefix) and k.endswith(suffix) and "fetch_bytes_corrected_per_launch" in v and "write_bytes_per_launch" in v:
            n += v["launches"]
            b += v["launches"] * (v["fetch_bytes_corrected_per_launch"] + v["write_bytes_per_launch"])
    return (round(b / n) if n else None), os.path.basename(files[-1])


def usable_cores():
    """(threads this process can actually run in parallel, how that was derived): the CPU affinity mask, cut down by the
    cgroup CPU quota when the container has one.  os.cpu_count() / torch's default say 128-256 on the GPU boxes, whose
    containers deliver a fraction of that (round 2 reported "128 cores" for a baseline that ran on far fewer)."""
    aff = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota, src = None, f"affinity {aff}"
    try:                                                  # cgroup v2
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = float(q) / float(p)
    except (OSError, ValueError):
        try:                                              # cgroup v1
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / p
        except (OSError, ValueError):
            pass
    n = aff
    if quota is not None:
        src += f", cgroup quota {quota:.1f} CPUs"
        n = max(1, min(aff, int(quota + 0.5)))
    return n, src


def cpu_baseline(model, seed, decoder, lm_path, budget_s=25.0, batch=64, clip_seconds=10.0):
    """Time the CPU oracle (port of the reference path on the same ATen CPU ops) on a bounded sample of the workload:
    best of up to 5 runs after one warm-up, inside ~25 s, on as many intra-op threads as the process can really use
    (usable_cores), which is what `cores` reports."""
    from oracle import quartznet_oracle as O   # checker / baseline only -- never on the product path
    cfg = configs.builtin(model)
    jas = cfg["JasperEncoder"]["jasper"]
    enc_sd = synth.encoder_state_dict(jas, 64, seed)
    dec_sd = synth.decoder_state_dict(jas[-1]["filters"], len(cfg["labels"]) + 1, seed)
    cores, how = usable_cores()
    prev_threads = torch.get_num_threads()
    torch.set_num_threads(cores)
    if decoder == "beam":
        from oracle import beam_oracle as BO
        b, seconds = 1, 2.0       # the restated pyctcdecode loop is pure Python: one 2 s utterance is ~10 s of CPU
        lm = BO.LanguageModel(BO.NgramLM.from_arpa(lm_path), alpha=0.5, beta=1.5, unigrams=BO.unigrams_for_path(lm_path)) if lm_path else None
    else:
        # 4 clips: the batch at which the CPU path is FASTEST per audio-second (measured on the GPU box's 16 cores: 292x real
        # time at 4 x 10 s, 77x at the workload's own 64 x 10 s, whose 67 MB activations per layer fall out of the caches) --
        # the baseline gets its best configuration, the whole-batch figure is reported beside it (`full_batch`)
        b, seconds = 4, 10.0
    sig, lens = synth.audio_batch(b, int(seconds * 16000), seed, ragged=False)

    def once():
        r = O.forward_all(sig, lens, enc_sd, dec_sd, jas)
        if decoder == "beam":
            return [BO.decode(r["logp"][i].numpy(), cfg["labels"], 128, lm=lm) for i in range(b)]
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
    full = None
    if decoder != "beam" and batch * clip_seconds <= 640.0 and batch > b:
        fs, fl = synth.audio_batch(batch, int(clip_seconds * 16000), seed, ragged=False)
        with torch.no_grad():
            t0 = time.perf_counter()
            O.ctc_decode_strings(O.forward_all(fs, fl, enc_sd, dec_sd, jas)["pred"], cfg["labels"])
            dt = time.perf_counter() - t0
        full = {"value": round(batch * clip_seconds / dt, 2), "sample": f"the workload's own batch, {batch} x {clip_seconds:g} s, one run"}
    torch.set_num_threads(prev_threads)
    best = min(times)
    return {"value": round(b * seconds / best, 2), "full_batch": full, "unit": "audio-sec/wall-sec", "cores": cores, "cores_how": how,
            "os_cpu_count": os.cpu_count(), "kind": "port",
            "sample": f"{model} {decoder}{' beam 128 + 3-gram LM' if decoder == 'beam' else ''}, batch {b} x {seconds:g} s, "
                      f"best of {len(times)} after 1 warm-up, {cores} intra-op threads"
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


_ENGINES = {}


def engine_for(model, dev, gemm, seed=3):
    key = (model, str(dev), gemm)
    if key not in _ENGINES:
        cfg = configs.builtin(model)
        jas = cfg["JasperEncoder"]["jasper"]
        enc_sd = synth.encoder_state_dict(jas, 64, seed)
        dec_sd = synth.decoder_state_dict(jas[-1]["filters"], len(cfg["labels"]) + 1, seed)
        _ENGINES[key] = (cfg, QuartzNetCTC(cfg, enc_sd, dec_sd, device=dev, gemm=gemm))
    return _ENGINES[key]


_LM = {}


def beam_decoder_for(cfg, no_lm, seed=3):
    from viet_asr_amd.beam import BeamSearchDecoder
    key = (bool(no_lm), len(cfg["labels"]))
    if key not in _LM:
        lm_path = None
        if not no_lm:        # every rank writes its own copy (deterministic, < 1 s)
            lm_path = os.path.join(tempfile.mkdtemp(prefix="vasr_lm_"), "synthetic3.arpa")
            synth.synthetic_arpa(lm_path, cfg["labels"], seed=seed)
        dec = BeamSearchDecoder(cfg["labels"], lm_path=lm_path, alpha=0.5, beta=1.5)
        lm = dec._get_lm()
        info = None
        if lm is not None:
            info = {"order": lm.order, "ngrams": lm.n_ngrams, "words": lm.n_words, "table_load": round(lm.table_load, 3),
                    # pyctcdecode's behaviour for a path ending in .arpa: unigram set + character trie (viet-asr_amd/beam.py)
                    "unigrams": "arpa" if lm.unigram_set is not None else "none", "trie_nodes": getattr(lm, "n_trie_nodes", 0)}
        _LM[key] = (dec, lm_path, info)
    return _LM[key]


def class_rates(prof, steps, gemm):
    """Per-class rates from vasr_profile_end -- work that ran / time it took, per class.  `roofline` is the DOMINANT kernel,
    the 1x1-conv GEMM (pw_gemm_split_kernel launches only); the fused depthwise + pointwise launches are a kernel of their
    own with two bounds (their GEMM against the MFMA peak, their traffic against HBM) and are reported under `fused`;
    fam_* = both together, the fused launches' whole duration charged to the GEMM (conservative); depthwise = the layers
    that ran as kernels of their own."""
    terms, optype, _ = GEMM_MODES[gemm]
    fu_ms = prof["fused"]["ms"] / steps
    pw_ms = prof["pointwise"]["ms"] / steps
    dw_ms = prof["depthwise"]["ms"] / steps
    pw_flops, fu_flops = prof["pointwise"]["flops"] / steps, prof["fused"]["flops"] / steps
    gemm_flops = pw_flops + fu_flops
    dw_bytes = prof["depthwise"]["bytes"] / steps
    peak = PEAK_F32_MFMA_TFLOPS if gemm == "fp32" else PEAK_16BIT_MFMA_TFLOPS
    pw_tflops = pw_flops / (pw_ms * 1e-3) / 1e12 if pw_ms else 0.0        # fp32-equivalent (algorithmic) rate
    fam_tflops = gemm_flops / ((pw_ms + fu_ms) * 1e-3) / 1e12 if pw_ms + fu_ms else 0.0
    dw_gbs = dw_bytes / (dw_ms * 1e-3) / 1e9 if dw_ms else 0.0
    return dict(terms=terms, peak=peak, fu_ms=fu_ms, pw_ms=pw_ms, dw_ms=dw_ms, gemm_flops=gemm_flops, pw_flops=pw_flops,
                fu_flops=fu_flops, dw_bytes=dw_bytes, pw_tflops=pw_tflops, exec_tflops=pw_tflops * terms,
                fam_exec_tflops=fam_tflops * terms, fam_ms=pw_ms + fu_ms, dw_gbs=dw_gbs,
                fu_exec_tflops=(fu_flops * terms / (fu_ms * 1e-3) / 1e12 if fu_ms else 0.0),
                fu_gbs=(prof["fused"]["bytes"] / steps / (fu_ms * 1e-3) / 1e9 if fu_ms else 0.0))


def side_workload(cfg_id, dev, gemm, steps, warmup, a, ragged=False):
    """One of the other BASELINE workloads, short: timed steps + the per-class pass; the compact record that goes under
    "configs" in the default line.  ragged: SURVEY section 8(d)'s ragged variant -- lengths uniform in [L/2, L], zero-padded to
    the longest row like the reference's collate (parts/dataset.py:14-53); value counts the REAL audio seconds."""
    model, batch, seconds, rate, decoder = WORKLOADS[cfg_id]
    cfg, eng = engine_for(model, dev, gemm)
    sig, lens = synth.audio_batch(batch, int(seconds * rate), 3, ragged=ragged)
    wav_in, ln_in = torch.from_numpy(sig).to(dev), torch.from_numpy(lens).to(dev)
    if rate != 16000:
        from viet_asr_amd import audio
    beam_dec, lm_info = None, None
    if decoder == "beam":
        beam_dec, _, lm_info = beam_decoder_for(cfg, a.no_lm)

    def inputs():
        return (wav_in, ln_in) if rate == 16000 else audio.resample(wav_in, ln_in, rate, 16000)

    def step():
        wav, ln = inputs()
        if decoder == "beam":
            return eng.forward_beam(wav, ln, beam_dec, a.beam_width, overlap=True)
        return eng.forward(wav, ln, want_logp=False, want_pred=False)

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        r = step()
    if r.get("done") is not None:
        r["done"].synchronize()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    wav16, ln16 = inputs()
    torch.cuda.synchronize()
    eng.handle.profile_begin()
    for _ in range(steps):
        eng.forward(wav16, ln16, want_logp=False, want_pred=False)
    torch.cuda.synchronize()
    cr = class_rates(eng.handle.profile_end(), steps, gemm)
    audio_s = float(lens.sum()) / rate
    out = {"workload": f"BASELINE configs[{cfg_id - 1}]: {model} {decoder}, batch={batch}x{seconds:g}s {rate // 1000}kHz"
                       + (" with ragged lengths U[L/2, L], zero-padded to the longest row" if ragged else ""),
           "steps": steps, "value": round(audio_s * steps / dt, 1), "ms_per_step": round(dt / steps * 1e3, 3),
           "utts_per_sec": round(batch * steps / dt, 1),
           "roofline": {"frac": round(cr["exec_tflops"] / cr["peak"], 4), "achieved": round(cr["exec_tflops"], 1),
                        "ms_per_step": round(cr["pw_ms"], 3),
                        "gemm_family_frac": round(cr["fam_exec_tflops"] / cr["peak"], 4), "gemm_family_ms_per_step": round(cr["fam_ms"], 3)},
           "depthwise": {"frac": round(cr["dw_gbs"] / PEAK_HBM_GBS, 4), "frac_of_achievable": round(cr["dw_gbs"] / ACHIEVABLE_HBM_GBS, 4),
                         "achieved": round(cr["dw_gbs"], 1), "ms_per_step": round(cr["dw_ms"], 3)},
           "fused_ms_per_step": round(cr["fu_ms"], 3)}
    if ragged:
        # padded-work efficiency of the padded batch the reference's collate builds: real frames / (rows x longest row)
        out["padded_work_efficiency"] = round(float(lens.sum()) / (batch * float(lens.max())), 4)
        out["value_note"] = "real audio-seconds (sum of the rows' own lengths) per wall-second; the kernels run the padded batch"
    if decoder == "beam":
        out["beam_width"], out["lm"] = a.beam_width, lm_info
        lp = eng.forward(wav16, ln16, want_logp=True, want_pred=False)["logp"]
        ms_beam = timed_ms(lambda: beam_dec.decode_ids(lp, a.beam_width), 3)
        ms_ac = timed_ms(lambda: eng.forward(wav16, ln16, want_logp=True, want_pred=False), 3)
        # a job's LAST batch has no following acoustic pass to hide its search under
        out["beam"] = {"search_ms_per_batch_alone": round(ms_beam, 3), "acoustic_ms_per_batch_alone": round(ms_ac, 3),
                       "last_batch_tail_ms": round(max(0.0, ms_beam + ms_ac - out["ms_per_step"]), 3)}
    del wav_in, wav16
    return out


def timed_ms(fn, n):
    """Mean milliseconds of n back-to-back calls on the current stream (one untimed call first)."""
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def sync_latency_ms(fn, n=30, warm=5):
    """Median wall milliseconds of one synchronised call (host launch overhead included)."""
    ts = []
    for i in range(n + warm):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        if i >= warm:
            ts.append(time.perf_counter() - t0)
    return round(float(np.median(ts)) * 1e3, 3)


def latency_block(dev, gemm, a):
    """Small-batch latency of the whole path (wav in HBM -> collapsed ids / best hypothesis on the device, synchronised per
    call).  The reference serves exactly this shape: batch 1, QuartzNet12x1 with the Vietnamese head, beam search with the
    3-gram LM at width 100 (CLI, infer.py:181-192) or 50 (app.py:22-28, 66-67) -- the `vi12x1_b1_*` entries, each split into
    the acoustic pass and the search.  Median of 30 calls after 5 warm-ups."""
    cfg, eng = engine_for("quartznet15x5", dev, gemm)
    out = {}
    for name, b, seconds in (("b1_10s_ms", 1, 10.0), ("b1_2s_ms", 1, 2.0), ("b8_10s_ms", 8, 10.0)):
        sig, lens = synth.audio_batch(b, int(seconds * 16000), 5, ragged=False)
        wav, ln = torch.from_numpy(sig).to(dev), torch.from_numpy(lens).to(dev)
        out[name] = sync_latency_ms(lambda: eng.forward(wav, ln, want_logp=False, want_pred=False))
    out["note"] = "quartznet15x5 greedy, one synchronised call per measurement (host launch overhead included), median of 30"
    # ---- the reference's own serving shape ----
    vcfg, veng = engine_for("quartznet12x1_vi", dev, gemm)
    dec, lm_path, lm_info = beam_decoder_for(vcfg, a.no_lm)
    words = None
    if lm_path:
        from viet_asr_amd.beam import read_arpa
        words = sorted(w[0] for w in read_arpa(lm_path)[1] if len(w) == 1 and not w[0].startswith("<"))
    vi = {"model": "quartznet12x1_vi (91 classes incl. blank), batch 1", "lm": lm_info,
          "note": "greedy_ms = wav -> collapsed ids; beamW_lm_ms = wav -> log-probs -> beam search (width W, 3-gram LM, alpha 0.5, "
                  "beta 1.5) -> best hypothesis, serial on one stream as a batch-1 server runs it; acoustic_ms = the log-prob "
                  "pass alone; search_ms = the search alone on this model's posteriors (a random-weight model is nearly "
                  "deterministic) and on CTC-like posteriors of the same shape that spell words of the LM (beams branch and "
                  "merge, the LM re-ranks at word boundaries)"}
    for seconds in (2.0, 6.6, 10.0):
        tag = f"vi12x1_b1_{seconds:g}s"
        sig, lens = synth.audio_batch(1, int(seconds * 16000), 5, ragged=False)
        wav, ln = torch.from_numpy(sig).to(dev), torch.from_numpy(lens).to(dev)
        e = {"greedy_ms": sync_latency_ms(lambda: veng.forward(wav, ln, want_logp=False, want_pred=False)),
             "acoustic_ms": sync_latency_ms(lambda: veng.forward(wav, ln, want_logp=True, want_pred=False))}
        lp = veng.forward(wav, ln, want_logp=True, want_pred=False)["logp"]
        lp_ctc = None
        if words:
            lp_ctc = torch.from_numpy(synth.ctc_like_log_probs(1, lp.shape[1], vcfg["labels"], words, seed=5)).to(dev)
        for width in (50, 100):
            e[f"beam{width}_lm_ms"] = sync_latency_ms(
                lambda: dec.decode_ids(veng.forward(wav, ln, want_logp=True, want_pred=False)["logp"], width))
            e[f"beam{width}_search_ms"] = sync_latency_ms(lambda: dec.decode_ids(lp, width))
            if lp_ctc is not None:
                e[f"beam{width}_search_ctc_like_ms"] = sync_latency_ms(lambda: dec.decode_ids(lp_ctc, width))
        vi[tag] = e
    out["vi12x1_b1"] = vi
    return out


def model_traffic_bytes(jas, feat_in, classes, batch, T):
    """Algorithmic HBM bytes of one pass of the encoder + head over [batch, feat_in, T] fp32, two ways:
    `unfused` -- every layer reads its input and writes its output (depthwise, 1x1 conv with the residual folded in as a
    second source, head), weights once; `fused` -- every depthwise + 1x1 sub-block as ONE kernel (the depthwise output
    never leaves the CU) and the CTC head + log-softmax + argmax folded into the last GEMM's consumer."""
    unf = fus = w = 0.0
    c, t = feat_in, T
    for blk in jas:
        k = blk["kernel"][0] if isinstance(blk["kernel"], (list, tuple)) else blk["kernel"]
        stride = blk["stride"][0] if isinstance(blk["stride"], (list, tuple)) else blk["stride"]
        cin_blk, t_blk = c, t
        for r in range(blk["repeat"]):
            last = r + 1 == blk["repeat"]
            res = blk.get("residual", False) and last
            to = (t - 1) // stride + 1 if stride > 1 else t
            if blk.get("separable", False):
                unf += 4.0 * batch * (c * t + c * to)                      # depthwise: read x, write d
                w += 4.0 * c * k
            unf += 4.0 * batch * (c * to + blk["filters"] * to + (cin_blk * t_blk if res else 0))   # 1x1 conv (+ residual source)
            fus += 4.0 * batch * (c * t + blk["filters"] * to + (cin_blk * t_blk if res else 0))
            w += 4.0 * c * blk["filters"] + (4.0 * cin_blk * blk["filters"] if res else 0)
            c, t = blk["filters"], to
    unf += 4.0 * batch * (c * t + classes * t) + 4.0 * batch * classes * t * 2     # head GEMM, log-softmax
    fus += 4.0 * batch * c * t + 8.0 * batch * t
    w += 4.0 * c * classes
    return unf + w, fus + w


def box_normalisers(dev, gemm="f16x2"):
    """What THIS box sustains, measured right after the timed region (chip warm) so that rounds on different boxes can be
    compared (three of four rounds' headline deltas were inside the box-to-box spread):
      * measured_mfma_tflops: the instruction stream of the GEMM in the arithmetic the line is quoted in (`gemm`), with
        everything but its MFMAs removed -- f16x2: three v_mfma_f32_32x32x16_f16 per tile pair and k-step on fp16 hi / lo
        planes; bf16x3: six bf16 ones on three planes -- 8 wavefronts per CU, the kernel's 2 x 4 tile order, operands with the
        bit statistics of scaled weights and rectified activations resident in registers (csrc/encoder_pw_split.hip
        mfma_sustained_kernel through the devtools build), 4 launches of ~8 ms.  Round 6: like for like (rounds 4-5 divided
        the f16x2 kernel by the bf16x3 stream); `measured_mfma_stream` names the stream;
      * measured_copy_gbs: read + write rate of a float4 streaming pass over 2 x 1 GiB (far beyond the 256 MiB Infinity
        Cache: HBM itself); measured_copy_gbs_cache_resident: the same over 2 x 67 MB -- the size of the headline workload's
        512-channel activations, which is what its depthwise layers are really bounded by.
    Measurement plumbing only (torch.clamp_min as the streaming pass): nothing here is on the product path."""
    import ctypes
    out = {}
    try:
        D = _lib.dev_lib()
        sink = torch.zeros(16, device=dev)
        fl = ctypes.c_double()
        st = torch.cuda.current_stream().cuda_stream
        n_cu = torch.cuda.get_device_properties(dev).multi_processor_count
        mode = {"bf16x3": 1, "f16x2": 3}.get(gemm, 3)       # (fp32 mode: no 16-bit stream; the f16 one is reported, unused)
        ksteps = 8000 if mode == 3 else 4000                # ~8 ms per launch either way (24 / 48 MFMAs per k-step)
        run = lambda: _lib.check(D.vasr_bench_mfma_sustained(mode, n_cu, ksteps, sink.data_ptr(), ctypes.byref(fl), st), D)
        run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4):
            run()
        e1.record()
        torch.cuda.synchronize()
        out["measured_mfma_tflops"] = round(4 * fl.value / (e0.elapsed_time(e1) * 1e-3) / 1e12, 1)
        out["measured_mfma_stream"] = {1: "bf16x3: 6 x v_mfma_f32_32x32x16_bf16 per tile pair and k-step",
                                       3: "f16x2: 3 x v_mfma_f32_32x32x16_f16 per tile pair and k-step"}[mode]
    except Exception as e:  # noqa: BLE001 -- a normaliser must not take the bench line down
        out["measured_mfma_error"] = repr(e)[:200]
    try:                    # (2 x 1 GiB after a 10 GB workspace -- an allocation failure here must not lose the line either)
        for key, n in (("measured_copy_gbs", 1 << 28), ("measured_copy_gbs_cache_resident", 64 * 512 * 512)):
            x = torch.randn(n, device=dev)
            y = torch.empty_like(x)
            for _ in range(3):
                torch.clamp_min(x, 0, out=y)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            reps = 10 if n > (1 << 26) else 50
            e0.record()
            for _ in range(reps):
                torch.clamp_min(x, 0, out=y)
            e1.record()
            torch.cuda.synchronize()
            out[key] = round(reps * 2 * 4 * n / (e0.elapsed_time(e1) * 1e-3) / 1e9, 1)
            del x, y
    except Exception as e:  # noqa: BLE001
        out["measured_copy_error"] = repr(e)[:200]
    torch.cuda.empty_cache()
    return out


def sol_block(cfg, work, batch, samples, step_ms, launches_b1, norm=None):
    """Speed-of-light table of the headline step, so that the measured step can be read against its floors without redoing
    the arithmetic.  MFMA floor: 3 fp16 products per multiply at the nominal 2.5 PFLOP/s and at the
    rate THIS box's bare MFMA stream sustains (power-limited: DESIGN section 4; round 4 used the constant 1.75 PFLOP/s here).
    HBM floors at the box's measured streaming rate (the guide's achievable 6.3 TB/s when the measurement failed)."""
    norm = norm or {}
    mfma = norm.get("measured_mfma_tflops") or 1750.0
    copy = norm.get("measured_copy_gbs") or 6300.0
    jas = cfg["JasperEncoder"]["jasper"]
    T = 1 + samples // 160
    unf, fus = model_traffic_bytes(jas, 64, len(cfg["labels"]) + 1, batch, T)
    unf += 4.0 * batch * samples + 4.0 * batch * 64 * T * 3        # front end: wav in, mel out, normalise read + write
    fus += 4.0 * batch * samples + 4.0 * batch * 64 * T
    exec_flops = 3.0 * (work["pointwise_flops"] + work["decoder_flops"])
    return {"step_ms_measured": round(step_ms, 3),
            "mfma_floor_ms_at_2500_tflops": round(exec_flops / 2.5e15 * 1e3, 3),
            "measured_mfma_tflops": norm.get("measured_mfma_tflops"), "measured_copy_gbs": norm.get("measured_copy_gbs"),
            "measured_copy_gbs_cache_resident": norm.get("measured_copy_gbs_cache_resident"),
            "mfma_floor_ms_at_measured_sustained": round(exec_flops / (mfma * 1e12) * 1e3, 3),
            "hbm_floor_ms_unfused_graph_at_measured_copy": round(unf / (copy * 1e9) * 1e3, 3),
            "hbm_floor_ms_fully_fused_graph_at_measured_copy": round(fus / (copy * 1e9) * 1e3, 3),
            "bytes_unfused_graph": unf, "bytes_fully_fused_graph": fus, "executed_mfma_flops": exec_flops,
            "b1_launches": launches_b1,
            "b1_launch_floor_ms": round(launches_b1 * 1.45e-3, 3) if launches_b1 else None,
            "step_over_mfma_floor_sustained": round(step_ms / (exec_flops / (mfma * 1e12) * 1e3), 2),
            "note": "measured_* = this box, right after the timed region (box_normalisers); floors are not additive (MFMA and HBM phases can overlap); b1_launch_floor = launches of a batch-1 call x the "
                    "1.45 us dependent-kernel boundary of MI355X_MICROARCH.md's price list"}


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
    ap.add_argument("--no-side-configs", action="store_true", help="skip the configs 2 / 4 / 5 and latency blocks of the default line")
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
    cfg, eng = engine_for(model, dev, a.gemm, seed)
    gemm = a.gemm or eng.handle.gemm_mode_name()
    _ENGINES[(model, str(dev), gemm)] = _ENGINES[(model, str(dev), a.gemm)]     # the side workloads ask by name
    samples = int(seconds * 16000)               # at the model's rate

    # ---- the batches this rank processes per step ----
    # Weak-scaling workloads: one batch of `batch` clips per rank and step.  The 4096-clip job (--config 5, N > 1): the
    # whole job per step, this rank's share in passes of <= `batch` clips; --ragged deals length buckets to the ranks
    # heaviest-first (dist.balanced_shards), otherwise every clip is 30 s and the share is contiguous.
    # (VASR_BENCH_FORCE_JOB=1: the job form on whatever ranks there are -- the GPU test runs it in a world of one)
    job = a.config == 5 and (world > 1 or bool(os.environ.get("VASR_BENCH_FORCE_JOB"))) and not a.model
    if job and a.batch is None:
        batch = min(batch, max(1, JOB_CLIPS // world))
    passes, sharding = [], None
    if job:
        from viet_asr_amd import dist as vdist
        if a.ragged:
            r = np.random.RandomState(seed)
            dur = r.uniform(seconds / 2, seconds, JOB_CLIPS)
            bucket = 64
            # passes of up to `batch` clips out of consecutive length buckets (similar lengths share a pass)
            groups, costs = vdist.job_passes(dur, rank, world, batch, bucket)
            for g in groups:
                n_in = int(max(dur[i] for i in g) * rate)
                sig, lens = synth.audio_batch(len(g), n_in, seed + 1000 * rank + len(passes), ragged=False)
                lens[:] = np.minimum(n_in, np.maximum(1, (dur[g] * rate).astype(np.int64)))
                for k in range(len(g)):
                    sig[k, lens[k]:] = 0
                passes.append((torch.from_numpy(sig).to(dev), torch.from_numpy(lens).to(dev), float(lens.sum()) / rate))
            sharding = {"kind": "duration-balanced length buckets (dist.balanced_shards)", "bucket": bucket,
                        "padded_work_imbalance": round((max(costs) - min(costs)) / max(costs), 4)}
        else:
            lo, hi = vdist.shard_range(JOB_CLIPS, rank, world)
            n_in = int(seconds * rate)
            for p0 in range(lo, hi, batch):
                nb = min(batch, hi - p0)
                sig, lens = synth.audio_batch(nb, n_in, seed + 100 * rank + p0, ragged=False)
                passes.append((torch.from_numpy(sig).to(dev), torch.from_numpy(lens).to(dev), float(lens.sum()) / rate))
            sharding = {"kind": "contiguous equal-count shards (every clip is 30 s)"}
    else:
        sig, lens = synth.audio_batch(batch, int(seconds * rate), seed + 100 * rank, ragged=a.ragged)
        passes.append((torch.from_numpy(sig).to(dev), torch.from_numpy(lens).to(dev), float(lens.sum()) / rate))
    audio_sec_per_step = sum(p[2] for p in passes)
    clips_per_step = sum(int(p[0].shape[0]) for p in passes)

    beam_dec, lm_path, lm_info = None, None, None
    if decoder == "beam":
        beam_dec, lm_path, lm_info = beam_decoder_for(cfg, a.no_lm, seed)
    if rate != 16000:
        from viet_asr_amd import audio

    def acoustic_input(p):
        if rate == 16000:
            return p[0], p[1]
        return audio.resample(p[0], p[1], rate, 16000)

    ring = None
    if dist is not None:
        from viet_asr_amd import dist as vdist
        ring = vdist.AsyncIdGather(world, dev)      # double-buffered async all_gather_into_tensor (viet-asr_amd/dist.py)

    def one_pass(p):
        wav, ln = acoustic_input(p)
        if decoder == "beam":
            r = eng.forward_beam(wav, ln, beam_dec, a.beam_width, overlap=not a.no_overlap)
        else:
            r = eng.forward(wav, ln, want_logp=False, want_pred=False)
        if ring is not None and not (job and a.ragged):
            # Result gather, one collective per returned tensor like actions.py:774-807, asynchronous and double-buffered
            if r.get("done") is not None:
                torch.cuda.current_stream().wait_event(r["done"])     # the gather reads what the side stream wrote
            ring.submit(r["ids"], r["id_len"])
        return r

    def step():
        r, parts = None, []
        for p in passes:
            r = one_pass(p)
            parts.append(r)
        if dist is not None and job and a.ragged:
            # ragged job: the ranks' passes differ in number and shape, so ONE gather per step of all of this rank's id
            # rows through the shape-exchanging gather of the product path (dist.gather_id_sequences: all_gather(shape)
            # + padded ids + lengths) -- the same number of collectives on every rank
            from viet_asr_amd import dist as vdist
            width = max(q["ids"].shape[1] for q in parts)
            ids = torch.cat([torch.nn.functional.pad(q["ids"], (0, width - q["ids"].shape[1])) for q in parts])
            vdist.gather_id_sequences(ids, torch.cat([q["id_len"] for q in parts]))
        return r

    def sync():
        if dist is not None:
            ring.drain()
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        r = step()
    torch.cuda.synchronize()
    own_elapsed = time.perf_counter() - t0        # this rank's own work, before it waits for the others
    sync()
    elapsed = time.perf_counter() - t0
    rccl = None
    if dist is not None:
        if not (job and a.ragged):
            g = ring.last()                # the gathered copy of this rank's last batch must be what the engine returned
            if not (torch.equal(g[0][rank], r["ids"]) and torch.equal(g[1][rank], r["id_len"])):
                raise RuntimeError("result gather returned something else than this rank's own shard at its index")
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        tot = torch.tensor([audio_sec_per_step, float(clips_per_step)], dtype=torch.float64, device=dev)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        audio_all, clips_all = float(tot[0].item()), int(round(float(tot[1].item())))
        ids_dev = torch.tensor([local], dtype=torch.int32, device=dev)
        all_dev = torch.empty((world,), dtype=torch.int32, device=dev)
        dist.all_gather_into_tensor(all_dev, ids_dev)
        own = torch.tensor([own_elapsed / a.steps * 1e3], dtype=torch.float64, device=dev)
        all_own = torch.empty((world,), dtype=torch.float64, device=dev)
        dist.all_gather_into_tensor(all_own, own)
        rccl = {"rccl_ranks": dist.get_world_size(), "backend": dist.get_backend(), "rank_devices": all_dev.tolist(),
                "rank_ms_per_step": {"min": round(float(all_own.min()), 3), "max": round(float(all_own.max()), 3),
                                     "note": "each rank's own kernels per step, before it waits for the other ranks"}}
    else:
        audio_all, clips_all = audio_sec_per_step, clips_per_step

    # ---- second pass, same K steps, with per-kernel-class HIP events on the launch stream ----
    wav16, ln16 = acoustic_input(passes[0])
    torch.cuda.synchronize()
    eng.handle.profile_begin()
    for _ in range(a.steps):
        eng.forward(wav16, ln16, want_logp=False, want_pred=False)
    torch.cuda.synchronize()
    prof = eng.handle.profile_end()
    # depthwise / pointwise / fused launches carry their own (start, stop) events (hipExtLaunchKernelGGL: the dispatch
    # packet's begin / end timestamps), so these are kernel durations as rocprofv3 --kernel-trace reports them
    # (profiles/rNN_bench_kernel_stats.csv); the 2-3 us dispatch gap between dependent launches is in ms_per_step only.
    work = eng.handle.algorithmic_work(int(passes[0][0].shape[0]), int(wav16.shape[1]))

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
        ms_step = elapsed / a.steps * 1e3
        wgs = int(_lib.lib().vasr_beam_workgroups(batch))
        group = wgs == batch and batch > 1
        extra["beam"] = {"kernel": ("beam_group_kernel<4> (an utterance on four wavefronts of one compute unit, merge table in LDS; the form for "
                                    "batches up to 64)" if group or batch == 1 else
                                    "beam_wave_kernel (one wavefront per utterance, four utterances per compute unit, merge table in LDS)"),
                         "bound": "instruction issue + LDS latency of one wavefront per SIMD (dependent round trips per frame)", "beam_width": a.beam_width,
                         "ms_per_batch_alone": round(ms_beam, 3), "acoustic_ms_per_batch_alone": round(ms_ac, 3),
                         "ms_per_batch_on_ctc_like_posteriors": round(ms_ctc, 3) if ms_ctc else None,
                         "serial_sum_ms": round(ms_beam + ms_ac, 3),
                         # a job's LAST batch has no following acoustic pass to hide its search under: it costs the
                         # serial sum, i.e. this much more than a steady-state step
                         "last_batch_tail_ms": round(max(0.0, ms_beam + ms_ac - ms_step), 3),
                         "workgroups": wgs, "cus": n_cu,
                         "cus_busy_frac": round(min(1.0, wgs / n_cu), 3), "overlapped": not a.no_overlap, "lm": lm_info}
    if rate != 16000 and rank == 0:
        ms_rs = timed(lambda: audio.resample(passes[0][0], passes[0][1], rate, 16000), max(3, a.steps // 4))
        in_b, out_b = passes[0][0].numel() * 4, wav16.numel() * 4
        extra["resample"] = {"kernel": "resample_up_kernel (integer up-sampling: both wings as one fp64 filter in LDS, 8 outputs "
                                       "per lane over a sliding register window)", "ms_per_batch": round(ms_rs, 3),
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
        tf = work["pointwise_flops"] / ((p2["pointwise"]["ms"] + p2["fused"]["ms"]) / a.steps * 1e-3) / 1e12
        other = {"gemm": other_mode, "value": round(audio_sec_per_step * a.steps / dt, 1),
                 "ms_per_step": round(dt / a.steps * 1e3, 3),
                 "pointwise_ms_per_step": round((p2["pointwise"]["ms"] + p2["fused"]["ms"]) / a.steps, 3),
                 "pointwise_fp32_equivalent_tflops": round(tf, 2)}
        eng.handle.set_gemm_mode(gemm)

    if rank == 0:
        if r.get("done") is not None:
            r["done"].synchronize()
        hyp = eng.texts(r["ids"], r["id_len"])
        cr = class_rates(prof, a.steps, gemm)
        assert abs(cr["gemm_flops"] - work["pointwise_flops"]) <= 1e-6 * work["pointwise_flops"], (cr["gemm_flops"], work["pointwise_flops"])
        terms, optype, dtype_str = GEMM_MODES[gemm]
        headline = a.config == 3 and not (a.model or a.batch or a.seconds or a.ragged)
        kname = "pw_gemm_kernel" if gemm == "fp32" else "pw_gemm_split_kernel"
        arith_tag = {"f16x2": ", 2>", "bf16x3": ", 0>", "bf16x2": ", 1>"}.get(gemm, "")
        shard5 = a.config == 5 and not job and not (a.model or a.batch or a.seconds or a.ragged)
        pmc_of = "" if headline else ("c5_" if shard5 else None)
        pw_traffic, traffic_src = pmc_traffic(kname, arith_tag, pmc_of) if pmc_of is not None else (None, None)
        dw_traffic, _ = pmc_traffic("dw_", "", pmc_of) if pmc_of is not None else (None, None)
        what = {"greedy": "greedy CTC", "beam": f"beam search (width {a.beam_width}" + (", 3-gram LM" if lm_info else ", no LM") + ")"}[decoder]
        if job:
            wl = (f"BASELINE configs[4]: the {JOB_CLIPS}-clip job, {model} {what}, {seconds:g}s {rate // 1000}kHz clips"
                  f"{' (lengths in [15, 30] s)' if a.ragged else ''} -> 16 kHz on the device, sharded over {world} ranks "
                  f"({clips_per_step} clips on rank 0 in {len(passes)} pass(es) of <= {batch}), wav in HBM -> collapsed ids")
        else:
            wl = (f"BASELINE configs[{a.config - 1}]: {model} {what}, batch={batch}x{seconds:g}s "
                  f"{rate // 1000}kHz mono per GPU{' (ragged lengths)' if a.ragged else ''}"
                  f"{' -> 16 kHz on the device' if rate != 16000 else ''}, wav in HBM -> "
                  f"{'collapsed ids' if decoder == 'greedy' else 'best-hypothesis ids'}")
        out = {
            "metric": "real_time_factor", "value": round(audio_all * a.steps / elapsed, 1),
            "unit": "audio-sec/wall-sec", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(elapsed / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong" if job else "weak",
            "vs_baseline": None, "dtype": dtype_str, "data": "synthetic",
            "config": {"workload": wl, "batch_per_gpu": batch, "clip_seconds": seconds, "clips_per_step_all_ranks": clips_all,
                       "parallelism": f"utterance-shard x{world}"},
            "utts_per_sec": round(clips_all * a.steps / elapsed, 1),
            "roofline": {"kernel": f"{kname} (1x1 conv as {'split-operand ' if terms > 1 else ''}MFMA GEMM + BN/residual/ReLU epilogue; "
                                   "the fused depthwise + pointwise launches are listed under `fused`, both together under gemm_family)",
                         "bound": "mfma", "achieved": round(cr["exec_tflops"], 2), "peak": cr["peak"],
                         "unit": "TFLOP/s", "frac": round(cr["exec_tflops"] / cr["peak"], 4),
                         "mfma_products_per_multiply": terms, "fp32_equivalent_tflops": round(cr["pw_tflops"], 2),
                         # the same algorithmic work against the matrix pipe's fp32 peak (what an fp32-MFMA kernel could
                         # reach at most): > 1 means the split arithmetic beats the best possible exact-fp32 GEMM
                         "frac_of_fp32_mfma_peak": round(cr["pw_tflops"] / PEAK_F32_MFMA_TFLOPS, 3),
                         "traffic": None, "traffic_offline": pw_traffic,
                         "traffic_note": "HBM bytes per launch from the committed PMC pass named in traffic_source (not measured in "
                                         "this run; null off the two profiled workloads: the headline and the configs[4] shard)", "traffic_source": traffic_src,
                         "flops_per_step": cr["pw_flops"], "ms_per_step": round(cr["pw_ms"], 3),
                         "launches_per_step": prof["pointwise"]["launches"] // a.steps,
                         # every 1x1-conv GEMM of the step, fused launches included with their WHOLE duration (depthwise
                         # producers and all) charged to the GEMM: the conservative family figure
                         "gemm_family": {"flops_per_step": cr["gemm_flops"], "ms_per_step": round(cr["fam_ms"], 3),
                                         "achieved": round(cr["fam_exec_tflops"], 2), "frac": round(cr["fam_exec_tflops"] / cr["peak"], 4),
                                         "launches_per_step": (prof["pointwise"]["launches"] + prof["fused"]["launches"]) // a.steps}},
            "depthwise": {"kernel": "depthwise conv kernels that ran as launches of their own (dw_toeplitz_kernel<K,DIL> on the matrix "
                                    "pipe; dw_conv_generic for the stride-2 prologue; dw_pair_kernel under --gemm fp32 | bf16x3 or "
                                    "VASR_DW_MFMA=0)", "bound": "hbm", "achieved": round(cr["dw_gbs"], 1),
                          "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(cr["dw_gbs"] / PEAK_HBM_GBS, 4),
                          # tensors of the headline workload (134 MB per layer) sit in the 256 MiB Infinity Cache, so
                          # `frac` there is cache-resident bandwidth; against what HBM itself sustains:
                          "achievable_peak": ACHIEVABLE_HBM_GBS, "frac_of_achievable": round(cr["dw_gbs"] / ACHIEVABLE_HBM_GBS, 4),
                          "traffic": None, "traffic_offline": dw_traffic, "bytes_per_step": cr["dw_bytes"],
                          "ms_per_step": round(cr["dw_ms"], 3), "launches_per_step": prof["depthwise"]["launches"] // a.steps},
            "fused": {"kernel": "dwpw_fused_kernel<K, DUAL> (depthwise + 1x1 conv + BN + residual + ReLU of a 256-channel sub-block in one "
                                "launch)", "launches_per_step": prof["fused"]["launches"] // a.steps,
                      "ms_per_step": round(cr["fu_ms"], 3), "flops_per_step": cr["fu_flops"],
                      "hbm_bytes_per_step": prof["fused"]["bytes"] / a.steps,
                      "bounds": "its GEMM against the MFMA peak, its traffic (read x + write y, the depthwise output never leaves "
                                "the CU) against HBM; the two-kernel form it replaces moves twice the bytes",
                      "achieved_GBps": round(cr["fu_gbs"], 1) if cr["fu_ms"] else None,
                      "hbm_frac": round(cr["fu_gbs"] / PEAK_HBM_GBS, 4) if cr["fu_ms"] else None,
                      "achieved_TFLOPs_executed": round(cr["fu_exec_tflops"], 1) if cr["fu_ms"] else None,
                      "mfma_frac": round(cr["fu_exec_tflops"] / cr["peak"], 4) if cr["fu_ms"] else None},
            "other_ms_per_step": {"frontend": round(prof["frontend"]["ms"] / a.steps, 3),
                                  "head": round(prof["head"]["ms"] / a.steps, 3)},
            "sample_transcript": hyp[0][:32],
        }
        if job and len(passes) > 1:
            out["roofline"]["note"] = out["depthwise"]["note"] = "kernel classes measured on the first pass of rank 0's share"
        if sharding:
            out["sharding"] = sharding
        out.update(extra)
        if rccl is not None:
            out.update(rccl)
        if other is not None:
            out["other_gemm_arithmetic"] = other
        norm = box_normalisers(dev, gemm)
        if norm.get("measured_mfma_tflops") and gemm != "fp32":
            out["roofline"]["measured_sustained_peak"] = norm["measured_mfma_tflops"]
            out["roofline"]["frac_of_measured"] = round(cr["exec_tflops"] / norm["measured_mfma_tflops"], 4)
            out["roofline"]["gemm_family"]["frac_of_measured"] = round(cr["fam_exec_tflops"] / norm["measured_mfma_tflops"], 4)
            if norm.get("measured_copy_gbs_cache_resident") and prof["pointwise"]["bytes"]:
                # A GEMM launch is two serial phases on every CU at once -- a main loop the matrix pipe bounds (at the rate the
                # box's power limit allows) and a store-only epilogue the memory system bounds -- so its floor on THIS box is
                # executed flops / measured MFMA rate + stored bytes / measured streaming rate (cache-resident: a layer's output
                # stays in the Infinity Cache for its consumer at these sizes).  Unlike frac_of_measured, whose MFMA-only
                # denominator follows the box's clock twice as closely as the kernel does, this fraction reproduces across
                # boxes (five boxes of round 5: 0.733-0.745).
                st_bytes = prof["pointwise"]["bytes"] / a.steps
                t_mfma = cr["pw_flops"] * cr["terms"] / (norm["measured_mfma_tflops"] * 1e12) * 1e3
                t_store = st_bytes / (norm["measured_copy_gbs_cache_resident"] * 1e9) * 1e3
                out["roofline"]["serial_floor"] = {"mfma_ms": round(t_mfma, 3), "store_ms": round(t_store, 3), "stored_bytes_per_step": st_bytes,
                                                   "frac": round((t_mfma + t_store) / cr["pw_ms"], 4),
                                                   "note": "(flops / box MFMA rate + stored bytes / box streaming rate) / measured GEMM time"}
        if norm.get("measured_copy_gbs"):
            out["depthwise"]["measured_copy_gbs"] = norm["measured_copy_gbs"]
            out["depthwise"]["measured_copy_gbs_cache_resident"] = norm["measured_copy_gbs_cache_resident"]
            out["depthwise"]["frac_of_measured_copy"] = round(cr["dw_gbs"] / norm["measured_copy_gbs"], 4)
            out["depthwise"]["frac_of_measured_copy_cache_resident"] = round(cr["dw_gbs"] / norm["measured_copy_gbs_cache_resident"], 4)
        out["box"] = norm
        if headline and world == 1 and not a.no_side_configs:
            side = {}
            for key, cid, rag in (("2", 2, False), ("3r", 3, True), ("4", 4, False), ("5", 5, False)):
                try:
                    side[key] = side_workload(cid, dev, gemm, 5, 2, a, ragged=rag)
                except Exception as e:  # noqa: BLE001 -- a side workload must not take the headline line down
                    side[key] = {"error": repr(e)[:200]}
                torch.cuda.empty_cache()
            out["configs"] = side
            try:
                out["latency"] = latency_block(dev, gemm, a)
            except Exception as e:  # noqa: BLE001
                out["latency"] = {"error": repr(e)[:300]}
            # launches of a batch-1 call of the headline model (per-class pass of one call)
            sig1, len1 = synth.audio_batch(1, samples, 5, ragged=False)
            w1, l1 = torch.from_numpy(sig1).to(dev), torch.from_numpy(len1).to(dev)
            eng.handle.profile_begin()
            eng.forward(w1, l1, want_logp=False, want_pred=False)
            torch.cuda.synchronize()
            p1 = eng.handle.profile_end()
            # front end = stft_logmel + normalize_chain (which carries seq and the length chain), head = GEMM + log-softmax /
            # argmax + collapse; the class counters come from the library, so only these five are counted by hand
            launches_b1 = sum(p1[k]["launches"] for k in ("depthwise", "pointwise", "fused")) + 2 + 3
            out["sol"] = sol_block(cfg, work, batch, samples, elapsed / a.steps * 1e3, launches_b1, norm)
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(model, seed, decoder, lm_path, batch=batch, clip_seconds=seconds)
        if dist is not None:
            import ctypes
            ctypes.CDLL(None).fflush(None)      # RCCL's version banner sits in C stdio's buffer: keep the JSON line last
        print(json.dumps(out, ensure_ascii=False), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
