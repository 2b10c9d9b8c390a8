"""-m gpu: the HIP path at BASELINE.json's OWN sizes -- configs[2] (15x5 greedy, 64 x 10 s), configs[3] (15x5 log-probs ->
beam 128 + n-gram LM, 64 x 501 frames) and one GPU's shard of configs[4] (512 x 30 s at 8 kHz -> 16 kHz -> 15x5 greedy).

Each test here checks (a) sampled rows against the oracle -- legitimate because full-length rows of a padded batch do not
depend on the other rows (quirk Q5 only touches rows shorter than the batch maximum), which the batch-invariance test pins
bit-for-bit -- and (b) size-independent properties on every row.  The WHOLE batches, every row and every frame against the
oracle with a recorded flip count, are in tests/test_gpu_flips.py (round 4: the oracle runs 64 x 10 s in 2-3 s on the GPU
box's host cores).  Reference wiring: /root/reference/infer.py:132-160 (DAG), :194-206 (CLI loop)."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT

pytestmark = pytest.mark.gpu

LOGP_REL = 2e-5          # of the largest |log-prob| (the synthetic 15x5 model reaches several hundred on 10 s clips)
LOGP_ABS = 5e-4


def _eng15(seed, gemm=None):
    from viet_asr_amd import configs, synth
    from viet_asr_amd.engine import QuartzNetCTC
    cfg = configs.builtin("quartznet15x5")
    jas = cfg["JasperEncoder"]["jasper"]
    enc_sd, dec_sd = synth.encoder_state_dict(jas, 64, seed), synth.decoder_state_dict(1024, 29, seed)
    return cfg, jas, enc_sd, dec_sd, QuartzNetCTC(cfg, enc_sd, dec_sd, gemm=gemm)


def _prefix_equal(a, b, n):
    """Compacted id rows are defined up to their length; what lies behind is scratch."""
    a, b, n = a.cpu().numpy(), b.cpu().numpy(), n.cpu().numpy()
    return all((a[i, : n[i]] == b[i, : n[i]]).all() for i in range(len(n)))


def _row_check(r, b, ref, tag):
    """GPU row b against an oracle run of that row alone: log-probs inside the stated tolerance, predictions identical
    wherever the oracle's top-2 margin is above fp32 round-off (1e-4 x the scale of the log-probs), transcripts equal."""
    lp, want = r["logp"][b].cpu(), ref["logp"][0]
    scale = float(want.abs().max())
    err = float((lp - want).abs().max())
    assert err <= LOGP_ABS + LOGP_REL * scale, (tag, b, err, scale)
    top2 = want.topk(2, -1).values
    clear = (top2[..., 0] - top2[..., 1]) > 1e-4 * max(1.0, scale / 10)
    assert (r["pred"][b].cpu()[clear] == ref["pred"][0][clear]).all(), (tag, b)
    assert float(clear.float().mean()) > 0.99
    assert float(r["enc_len"][b]) == float(ref["enc_len"][0])


def test_config3_quartznet15x5_b64_10s_full_size(gpu):
    """BASELINE configs[2] at its own size, default GEMM arithmetic: three sampled full-length rows against the oracle,
    properties on all 64."""
    from viet_asr_amd import stages, synth
    from oracle import quartznet_oracle as O
    cfg, jas, enc_sd, dec_sd, eng = _eng15(3)
    sig, lens = synth.audio_batch(64, 160000, 3)
    wav, ln = torch.from_numpy(sig).to(gpu), torch.from_numpy(lens).to(gpu)
    r = eng.forward(wav, ln, want_logp=True)
    r2 = eng.forward(wav, ln, want_logp=True)
    assert torch.equal(r["logp"], r2["logp"]) and torch.equal(r["pred"], r2["pred"])          # deterministic
    assert torch.equal(r["id_len"], r2["id_len"]) and _prefix_equal(r["ids"], r2["ids"], r["id_len"])
    assert r["logp"].shape == (64, 501, 29) and bool(torch.isfinite(r["logp"]).all())
    assert float(torch.logsumexp(r["logp"].double(), -1).abs().max()) < 1e-3                  # rows are log-distributions
    assert torch.equal(r["logp"].argmax(-1), r["pred"])
    assert r["enc_len"].tolist() == [500.0] * 64          # 160000 % hop == 0: T = seq + 1 = 1001 mel frames, seq = 1000 (quirk Q2)
    for b in (0, 31, 63):
        ref = O.forward_all(sig[b:b + 1], lens[b:b + 1], enc_sd, dec_sd, jas)
        _row_check(r, b, ref, "config3")
        assert eng.texts(r["ids"][b:b + 1], r["id_len"][b:b + 1]) == O.ctc_decode_strings(ref["pred"], cfg["labels"])
    # the collapse of every row equals the host-side rule on that row's predictions (helpers.py:7-33)
    pred = r["pred"].cpu().numpy()
    ids, n = r["ids"].cpu().numpy(), r["id_len"].cpu().numpy()
    for b in range(64):
        assert ids[b, : n[b]].tolist() == O.ctc_collapse_ids(pred[b], 28)
    ids2, n2 = stages.ctc_collapse(r["pred"], 28)
    assert torch.equal(n2, r["id_len"]) and _prefix_equal(ids2, r["ids"], n2)


def test_config2_quartznet12x1_vi_b32_10s_full_size(gpu):
    """BASELINE configs[1] at its own size (QuartzNet12x1, Vietnamese head of 91 classes, 32 x 10 s), default GEMM
    arithmetic: three sampled full-length rows against the oracle -- round 2 checked this size through properties only
    -- and the properties on all 32."""
    from viet_asr_amd import configs, synth
    from viet_asr_amd.engine import QuartzNetCTC
    from oracle import quartznet_oracle as O
    cfg = configs.builtin("quartznet12x1_vi")
    jas = cfg["JasperEncoder"]["jasper"]
    enc_sd, dec_sd = synth.encoder_state_dict(jas, 64, 2), synth.decoder_state_dict(1024, 91, 2)
    eng = QuartzNetCTC(cfg, enc_sd, dec_sd)
    sig, lens = synth.audio_batch(32, 160000, 2)
    wav, ln = torch.from_numpy(sig).to(gpu), torch.from_numpy(lens).to(gpu)
    r = eng.forward(wav, ln, want_logp=True)
    eng.handle.profile_begin()
    r2 = eng.forward(wav, ln, want_logp=True)
    torch.cuda.synchronize()
    # round 4: the six 256-channel sub-blocks (each with its folded residual) are ONE kernel each here as well -- 32 x 4 tiles
    # of 128 frames would fill half the chip, the kernel's 64-frame form fills it
    assert eng.handle.profile_end()["fused"]["launches"] == 6
    assert torch.equal(r["logp"], r2["logp"]) and torch.equal(r["pred"], r2["pred"])
    assert r["logp"].shape == (32, 501, 91) and bool(torch.isfinite(r["logp"]).all())
    assert float(torch.logsumexp(r["logp"].double(), -1).abs().max()) < 1e-3
    assert torch.equal(r["logp"].argmax(-1), r["pred"])
    assert r["enc_len"].tolist() == [500.0] * 32
    for b in (0, 13, 31):
        ref = O.forward_all(sig[b:b + 1], lens[b:b + 1], enc_sd, dec_sd, jas)
        _row_check(r, b, ref, "config2")
        assert eng.texts(r["ids"][b:b + 1], r["id_len"][b:b + 1]) == O.ctc_decode_strings(ref["pred"], cfg["labels"])
    pred = r["pred"].cpu().numpy()
    ids, n = r["ids"].cpu().numpy(), r["id_len"].cpu().numpy()
    for b in range(32):
        assert ids[b, : n[b]].tolist() == O.ctc_collapse_ids(pred[b], 90)


def _beam_rows_match_oracle(texts, score, logp, rows, labels, olm, tag):
    from oracle import beam_oracle as BO
    excused, worst = 0, 0.0
    for b in rows:
        ref = BO.decode_beams(np.exp(logp[b].double().cpu().numpy()), labels, 128, lm=olm)
        # near-ties between the two best hypotheses may legitimately resolve differently (fp rounding), as in test_beam.py
        close = len(ref) > 1 and abs(ref[0][2] - ref[1][2]) < 1e-3
        assert texts[b] == ref[0][0] or (close and texts[b] == ref[1][0]), (tag, b, texts[b][:80], ref[0][0][:80])
        if texts[b] == ref[0][0]:
            assert abs(float(score[b]) - ref[0][2]) < 2e-3 * max(1.0, abs(ref[0][2]) / 50), (tag, b, float(score[b]), ref[0][2])
            worst = max(worst, abs(float(score[b]) - ref[0][2]))
        else:
            excused += 1
    from test_gpu_parity import _record
    _record("config4_beam_rows", tag=tag, rows=len(list(rows)), decided_by_the_near_tie_allowance=excused, worst_score_err=worst)


def test_config4_quartznet15x5_beam128_lm_b64(gpu, tmp_path):
    """BASELINE configs[3]: the 15x5 log-probs of 64 x 10 s -> device beam search, beam_width 128, 3-gram ARPA model of
    ~1.2e5 n-grams (realistic hash-table load; the reference's KenLM files are absent).  Rows are compared with the
    restated pyctcdecode (oracle/beam_oracle.py -- parity UNPINNED: third-party algorithm) fed with the SAME log-probs:
      (a) the model's own log-probs, as the reference wires it (infer.py:146-160) -- a random-weight model is nearly
          deterministic (~1.1 classes per frame above token_min_logp), so this checks the wiring at size;
      (b) CTC-like log-probs of the same shape spelling words of the LM's vocabulary (synth.ctc_like_log_probs), where
          beams branch, merge and are re-ranked by the LM at word boundaries: the LM must change at least one
          transcript, and scores must agree with the oracle.
    Every row: the oracle's transcript and score (round 5: all 128 searched rows, not a sample), determinism, label range,
    normalised spacing, finite score, and the same bits from the four-wavefront kernel as from the one-wavefront kernel."""
    from viet_asr_amd import synth
    from viet_asr_amd.beam import BeamSearchDecoder
    from oracle import beam_oracle as BO
    cfg, jas, enc_sd, dec_sd, eng = _eng15(3)
    labels = cfg["labels"]
    sig, lens = synth.audio_batch(64, 160000, 3)
    wav, ln = torch.from_numpy(sig).to(gpu), torch.from_numpy(lens).to(gpu)
    logp_model = eng.forward(wav, ln, want_logp=True)["logp"]
    assert logp_model.shape == (64, 501, 29)
    arpa = str(tmp_path / "synthetic3.arpa")
    ng = synth.synthetic_arpa(arpa, labels, seed=3)
    assert len(ng) > 100000
    words = sorted(w[0] for w in ng if len(w) == 1 and not w[0].startswith("<"))
    logp_ctc = torch.from_numpy(synth.ctc_like_log_probs(64, 501, labels, words, seed=3)).to(gpu)
    nolm = BeamSearchDecoder(labels, lm_path=None)
    # round 6: BOTH of pyctcdecode's LM behaviours (oracle/beam_oracle.py header) -- "arpa" = what build_ctcdecoder does for the
    # .arpa path handed over here (unigram set + character trie: the default), "binary" = no unigram list
    texts_by_mode = {}
    for mode in ("arpa", "binary"):
        dec = BeamSearchDecoder(labels, lm_path=arpa, alpha=0.5, beta=1.5, unigrams="auto" if mode == "arpa" else None)
        olm = BO.LanguageModel(BO.NgramLM.from_arpa(arpa), alpha=0.5, beta=1.5, unigrams=BO.unigrams_for_path(arpa) if mode == "arpa" else None)
        # (round 5: EVERY row against the Python oracle -- 0.1 s per row -- where rounds 2-4 sampled five of the 128)
        for tag, logp, rows in (("model", logp_model, range(64)), ("ctc-like", logp_ctc, range(64))):
            ids, n, score = dec.decode_ids(logp, 128)
            ids_b, n_b, score_b = dec.decode_ids(logp, 128)
            assert torch.equal(n, n_b) and torch.equal(score, score_b) and _prefix_equal(ids, ids_b, n), tag   # deterministic
            ids, n, score = ids.cpu().numpy(), n.cpu().numpy(), score.cpu().numpy()
            assert np.isfinite(score).all()
            texts = ["".join(labels[c] for c in ids[b, : n[b]]) for b in range(64)]
            for b, t in enumerate(texts):
                assert (ids[b, : n[b]] >= 0).all() and (ids[b, : n[b]] < 28).all()
                assert "  " not in t and t == t.strip(), (tag, b, t[:60])
            _beam_rows_match_oracle(texts, score, logp, rows, labels, olm, tag + "/" + mode)
            texts_by_mode[(tag, mode)] = texts
            # ALL 64 rows cross-checked between the two independent kernel forms: the batch of 64 went through beam_group.hip (an
            # utterance on four wavefronts; round 6: the form for batches up to 64), the same rows twice over = 128 rows go through
            # beam_wave.hip (one wavefront per utterance, half the pairs per pass): hypotheses, lengths and scores must be the
            # same bits (the Python oracle covers the rows above)
            ids_w, n_w, score_w = dec.decode_ids(torch.cat([logp, logp]), 128)
            n_w, score_w, ids_w = n_w.cpu().numpy(), score_w.cpu().numpy(), ids_w.cpu().numpy()
            for half in (0, 64):
                assert np.array_equal(n_w[half:half + 64], n) and np.array_equal(score_w[half:half + 64], score), (tag, half)
                for r_ in range(64):
                    assert np.array_equal(ids_w[half + r_, : n[r_]], ids[r_, : n[r_]]), (tag, half, r_)
            if tag == "ctc-like":
                plain = nolm.decode_batch(logp, 128)
                changed = sum(a != b for a, b in zip(plain, texts))
                assert changed >= 1, "the language model never changed a transcript: it is not being exercised"
                wset = set(words)
                in_vocab = lambda t: np.mean([w in wset for w in t.split()] or [0])
                assert np.mean([in_vocab(t) for t in texts]) >= np.mean([in_vocab(t) for t in plain])
    # the two behaviours are not the same search: on word-spelling posteriors the trie keeps partial words of the vocabulary alive
    assert texts_by_mode[("ctc-like", "arpa")] != texts_by_mode[("ctc-like", "binary")]
    from test_gpu_parity import _record
    _record("config4_lm_modes", rows_that_differ={t: sum(a != b for a, b in zip(texts_by_mode[(t, "arpa")], texts_by_mode[(t, "binary")]))
                                                   for t in ("model", "ctc-like")})
    dec = BeamSearchDecoder(labels, lm_path=arpa, alpha=0.5, beta=1.5)
    lm = dec._get_lm()
    assert lm.n_ngrams == len(ng) and 0.05 < lm.table_load < 0.6
    # batched + overlapped form (search of batch k on a side stream): same answer as the serial call
    ids, n, score = dec.decode_ids(logp_model, 128)
    out = eng.forward_beam(wav, ln, dec, 128, overlap=True)
    out["done"].synchronize()
    assert torch.equal(out["id_len"], n) and _prefix_equal(out["ids"], ids, n)


def test_config5_shard_512x30s_8khz(gpu):
    """One GPU's shard of BASELINE configs[4]: 512 clips of 30 s at 8 kHz, resampled to 16 kHz on the device, through
    QuartzNet15x5 greedy in ONE pass (10 GB of workspace).  Two full-length rows against the oracle chain (oracle
    resampler -> oracle model); every row: shapes, finiteness, log-distribution rows, collapse rule."""
    from viet_asr_amd import audio, synth
    from oracle import audio_oracle as AO
    from oracle import quartznet_oracle as O
    cfg, jas, enc_sd, dec_sd, eng = _eng15(5)
    B = 512
    sig, lens = synth.audio_batch(B, 240000, 5, ragged=True)
    lens[[0, 300]] = 240000                                  # the sampled rows are full length (independent of the rest)
    sig8, l8 = torch.from_numpy(sig).to(gpu), torch.from_numpy(lens).to(gpu)
    x16, l16 = audio.resample(sig8, l8, 8000, 16000)
    assert x16.shape == (B, 480000) and l16.tolist() == (2 * lens).tolist()
    r = eng.forward(x16, l16, want_logp=True)
    torch.cuda.synchronize()
    assert r["logp"].shape == (B, 1501, 29) and bool(torch.isfinite(r["logp"]).all())
    assert float(torch.logsumexp(r["logp"].double(), -1).abs().max()) < 1e-3
    assert torch.equal(r["logp"].argmax(-1), r["pred"])
    want_len = [float((int(np.ceil(2 * l / 160)) - 1) // 2 + 1) for l in lens]
    assert r["enc_len"].tolist() == want_len
    for b in (0, 300):
        up = AO.resample(sig[b, : lens[b]], 8000, 16000)
        assert float(np.abs(x16[b, : len(up)].cpu().numpy() - up).max()) <= 2e-6
        ref = O.forward_all(up[None], np.array([len(up)]), enc_sd, dec_sd, jas)
        _row_check(r, b, ref, "config5")
    pred = r["pred"].cpu().numpy()
    ids, n = r["ids"].cpu().numpy(), r["id_len"].cpu().numpy()
    for b in range(0, B, 37):
        assert ids[b, : n[b]].tolist() == O.ctc_collapse_ids(pred[b], 28)
    del r, x16
    torch.cuda.empty_cache()


_NCCL_SNIPPET = r"""
import os, sys
sys.path.insert(0, {root!r})
import numpy as np, torch, torch.distributed as dist
import viet_asr_amd
from viet_asr_amd import configs, synth, dist as vdist
from viet_asr_amd.engine import QuartzNetCTC
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29671", RANK="0", WORLD_SIZE="1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", init_method="env://", device_id=torch.device("cuda", 0))    # RCCL, world of one
cfg = configs.builtin("quartznet12x1_vi"); jas = cfg["JasperEncoder"]["jasper"]
eng = QuartzNetCTC(cfg, synth.encoder_state_dict(jas, 64, 2), synth.decoder_state_dict(1024, 91, 2))
sig, lens = synth.audio_batch(5, 24000, 2, ragged=True)
utts = [sig[b, : lens[b]] for b in range(5)]
got = vdist.transcribe_sharded(eng, utts)
want = eng.transcribe(utts)
ids = torch.arange(12, dtype=torch.int32, device="cuda").reshape(3, 4); n = torch.tensor([4, 2, 1], dtype=torch.int32, device="cuda")
a, b = vdist.gather_id_sequences(ids, n)
ok = got == want and torch.equal(a, ids) and torch.equal(b, n)
t = torch.ones(3, device="cuda"); dist.all_reduce(t)                                           # a real RCCL collective
ok = ok and t.tolist() == [1.0, 1.0, 1.0]
dist.destroy_process_group()
print("NCCL_OK" if ok else "NCCL_BAD %r %r" % (got, want))
"""


def test_rccl_world_of_one_transcribe_sharded(gpu):
    """dist.transcribe_sharded + gather_id_sequences over the nccl (= RCCL) backend, world size 1, in a child process
    (a process group cannot be re-created inside the pytest process)."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    out = subprocess.run([sys.executable, "-c", _NCCL_SNIPPET.format(root=ROOT)], env=env, capture_output=True, text=True,
                         timeout=600)
    assert "NCCL_OK" in out.stdout, (out.stdout[-2000:], out.stderr[-2000:])


def _bench(args, timeout=900):
    import json
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=timeout)
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    return out.returncode, (json.loads(lines[-1]) if lines else None), out.stdout[-1500:] + out.stderr[-1500:]


def test_bench_self_launch_matches_plain_run_and_refuses_missing_devices(gpu):
    """`python bench.py --gpus N` starts its own ranks.  --spawn forces that path at N = 1 (RCCL world of one, the result
    gather in the loop): its value must equal the plain single-process run within 2 % (best of two each -- the two forms
    run the same kernels; the gather is asynchronous).  With more GPUs requested than the box has: a clear message and
    exit code 3, nothing run."""
    common = ["--steps", "30", "--warmup", "5", "--no-cpu-baseline", "--no-other-gemm", "--no-side-configs"]
    plain, spawned = [], []
    for _ in range(2):
        rc, j, log = _bench(common)
        assert rc == 0 and j is not None, log
        plain.append(j["ms_per_step"])
        rc, j, log = _bench(common + ["--spawn"])
        assert rc == 0 and j is not None, log
        assert j["rccl_ranks"] == 1 and j["rank_devices"] == [0] and j["backend"] == "nccl"
        spawned.append(j["ms_per_step"])
    assert abs(min(spawned) / min(plain) - 1.0) <= 0.02, (plain, spawned)
    have = torch.cuda.device_count()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(have + 1)], capture_output=True, text=True,
                         timeout=300)
    assert out.returncode == 3 and f"needs {have + 1} visible HIP devices" in out.stderr, (out.returncode, out.stderr[-500:])


def test_bench_default_line_carries_every_baseline_config_and_the_latency_block(gpu):
    """the driver only ever sees the default line, so it must carry configs 2, 4 and 5 (5 steps each) and the
    batch-1 / batch-8 latency of the whole path next to the headline, with their roofline fractions."""
    rc, j, log = _bench(["--steps", "5", "--warmup", "2", "--no-cpu-baseline", "--no-other-gemm"])
    assert rc == 0 and j is not None, log
    assert "configs[2]" in j["config"]["workload"] and j["value"] > 2000
    for cid, tag in (("2", "quartznet12x1_vi"), ("4", "beam"), ("5", "512x30s")):
        c = j["configs"][cid]
        assert "error" not in c, c
        assert tag in c["workload"] and c["steps"] == 5 and c["value"] > 2000 and c["ms_per_step"] > 0
        assert 0 < c["roofline"]["frac"] < 1 and 0 < c["depthwise"]["frac"] < 1 and 0 < c["depthwise"]["frac_of_achievable"] < 1.3
    assert j["configs"]["4"]["lm"]["ngrams"] > 100000
    lat = j["latency"]
    assert 0 < lat["b1_2s_ms"] <= lat["b1_10s_ms"] <= lat["b8_10s_ms"] < 20
    assert j["depthwise"]["frac_of_achievable"] > j["depthwise"]["frac"]
    assert j["fused"]["launches_per_step"] == 30        # the 256-channel sub-blocks of 15x5 run as one kernel each


def test_bench_job_mode_of_config5_in_a_world_of_one(gpu):
    """`--gpus N --config 5` walks the whole 4096-clip job (strong scaling, per-rank min / max); the code path is run here
    with the job shrunk to 256 clips on the one GPU there is (RCCL world of one via --spawn), equal-length and ragged."""
    import json
    env = dict(os.environ, VASR_BENCH_FORCE_JOB="1", VASR_BENCH_JOB_CLIPS="256")
    for extra in ([], ["--ragged"]):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", "5", "--spawn", "--batch", "128", "--steps",
                              "2", "--warmup", "1", "--no-cpu-baseline", "--no-other-gemm"] + extra, env=env, capture_output=True,
                             text=True, timeout=900)
        lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
        assert out.returncode == 0 and lines, out.stdout[-1500:] + out.stderr[-1500:]
        j = json.loads(lines[-1])
        assert j["scaling"] == "strong" and j["config"]["clips_per_step_all_ranks"] == 256 and j["rccl_ranks"] == 1
        assert "256-clip job" in j["config"]["workload"] and j["value"] > 2000
        assert j["rank_ms_per_step"]["min"] == j["rank_ms_per_step"]["max"] > 0
        assert ("padded_work_imbalance" in j["sharding"]) == bool(extra)


@pytest.mark.parametrize("config", [2, 4, 5])
def test_bench_config_modes_print_the_contract(gpu, config):
    """bench.py --config {2,4,5}: same JSON contract as the default line, workload named after BASELINE.json's entry."""
    rc, j, log = _bench(["--config", str(config), "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-other-gemm"])
    assert rc == 0 and j is not None, log
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline"):
        assert k in j, k
    assert f"configs[{config - 1}]" in j["config"]["workload"] and j["value"] > 2000
    if config == 4:
        assert j["beam"]["beam_width"] == 128 and j["beam"]["lm"]["ngrams"] > 100000 and j["beam"]["workgroups"] == 64   # (round 6: up to 64 utterances a compute unit each)
    if config == 5:
        assert j["config"]["batch_per_gpu"] == 512 and j["resample"]["ms_per_batch"] > 0
