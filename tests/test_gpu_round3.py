"""-m gpu: parity cases added in round 3 (VERDICT r02 "next round" item 1 and ADVICE r02).

  * stft_conv=True (configs/quartznet15x5.yaml:26 -> third-party torch_stft, PARITY UNPINNED) against the oracle's
    restatement of that package's transform;
  * the default 2 x fp16 split arithmetic end to end on signals chosen to stress a per-utterance scale: a full-scale
    click in -80 dB noise, a DC offset of 0.5, hard clipping -- against the oracle AND against the exact-fp32 GEMM mode,
    same tolerance as the goldens;
  * ragged batches in row-independent mode: a row's result is bit-identical whatever batch it sits in (f16x2 included:
    the CTC head's operand scale now comes from the row's own frames);
  * beam search: the LM score cache's effect on the final </s> pass, which round 2's kernel did not model.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from test_gpu_parity import LOGP_REL, LOGP_TOL, MEL_TOL, _record, logp_tol  # noqa: E402


def test_stft_conv_frontend_matches_the_torch_stft_restatement(gpu):
    """AudioToMelSpectrogramPreprocessor(stft_conv=True): the kernels run the same centred DFT with torch_stft's
    periodic window; compared with oracle.torch_stft_magnitude -> features.py:260-301.  The two window conventions
    differ by far more than the tolerance, so the test also shows that the switch does something."""
    from viet_asr_amd import _lib, configs, stages, synth
    from viet_asr_amd.frontend_tables import frontend_description
    from oracle import quartznet_oracle as O
    cfg = configs.builtin("quartznet15x5")
    pre = dict(cfg["AudioToMelSpectrogramPreprocessor"], stft_conv=True)
    sig, lens = synth.audio_batch(4, 40000, 31, ragged=True)
    lens[2] = 160 * 101
    sig[2, lens[2]:] = 0
    h = _lib.Handle(frontend=frontend_description(pre)); h.finalize()
    mel, seq = stages.melspec(h, torch.from_numpy(sig).to(gpu), torch.from_numpy(lens).to(gpu))
    ref, ref_seq = O.melspec_forward(sig, lens, stft_conv=True)
    plain, _ = O.melspec_forward(sig, lens, stft_conv=False)
    err = float((mel.cpu() - ref).abs().max())
    _record("stft_conv", err=err, window_effect=float((plain - ref).abs().max()))
    assert (seq.cpu() == ref_seq).all()
    assert err <= MEL_TOL, err
    assert float((plain - ref).abs().max()) > 50 * MEL_TOL


def _adversarial_batch(n):
    r = np.random.RandomState(1234)
    sig = np.zeros((3, n), dtype=np.float32)
    sig[0] = 1e-4 * r.randn(n)                         # -80 dB noise floor ...
    sig[0, n // 3] = 1.0                               # ... with one full-scale click
    sig[0, 2 * n // 3: 2 * n // 3 + 3] = [-1.0, 1.0, -1.0]
    sig[1] = 0.5 + 0.05 * r.randn(n)                   # DC offset 0.5
    sig[2] = np.clip(2.5 * r.randn(n), -1.0, 1.0)      # hard clipping: most samples sit on the rails
    lens = np.array([n, n - 1234, n - 4321], dtype=np.int64)
    for b in range(3):
        sig[b, lens[b]:] = 0
    return sig, lens


@pytest.mark.parametrize("model", ["quartznet15x5", "quartznet12x1_vi"])
def test_f16x2_end_to_end_on_adversarial_signals(gpu, model):
    from viet_asr_amd import configs, synth
    from viet_asr_amd.engine import QuartzNetCTC
    from oracle import quartznet_oracle as O
    cfg = configs.builtin(model)
    jas = cfg["JasperEncoder"]["jasper"]
    enc_sd = synth.encoder_state_dict(jas, 64, 77)
    dec_sd = synth.decoder_state_dict(1024, len(cfg["labels"]) + 1, 77)
    sig, lens = _adversarial_batch(48000)
    ref = O.forward_all(sig, lens, enc_sd, dec_sd, jas)
    want = ref["logp"].numpy()
    tol = logp_tol(want)
    top2 = ref["logp"].topk(2, -1).values
    clear = ((top2[..., 0] - top2[..., 1]) > 2 * tol).numpy()
    out = {}
    for gemm in ("f16x2", "fp32"):
        eng = QuartzNetCTC(cfg, enc_sd, dec_sd, gemm=gemm)
        r = eng.forward(torch.from_numpy(sig).to(gpu), torch.from_numpy(lens).to(gpu), want_logp=True)
        torch.cuda.synchronize()
        got = r["logp"].cpu().numpy()
        assert np.isfinite(got).all()
        err = np.abs(got - want).max(axis=(1, 2))
        _record("adversarial", model=model, gemm=gemm, err_click=err[0], err_dc=err[1], err_clip=err[2], tol=tol,
                scale=np.abs(want).max())
        assert (err <= tol).all(), (gemm, err, tol)
        pred = r["pred"].cpu().numpy()
        assert (pred[clear] == ref["pred"].numpy()[clear]).all(), gemm
        assert clear.mean() > 0.9       # (padded frames of the shorter rows carry near-ties)
        assert (r["enc_len"].cpu().numpy() == ref["enc_len"].numpy()).all()
        out[gemm] = (got, eng.texts(r["ids"], r["id_len"]))
    # the split arithmetic is as close to the reference as exact fp32 MFMA is (same bound), and they agree with each other
    assert np.abs(out["f16x2"][0] - out["fp32"][0]).max() <= 2 * tol
    if clear.all():
        assert out["f16x2"][1] == out["fp32"][1] == O.ctc_decode_strings(ref["pred"], cfg["labels"])


@pytest.mark.parametrize("gemm", ["f16x2", "bf16x3"])
def test_row_independent_ragged_batches_are_bit_identical_per_row(gpu, gemm):
    """in f16x2 mode the CTC head's per-utterance scale used to come from every column below the BATCH's
    T' -- padded frames included -- so a row's bits could depend on its neighbours' lengths.  Row-independent mode now
    takes the maxima over the row's own frames and zeroes the head's input behind them."""
    from viet_asr_amd import configs, synth
    from viet_asr_amd.engine import QuartzNetCTC
    cfg = configs.builtin("quartznet12x1_vi")
    jas = cfg["JasperEncoder"]["jasper"]
    enc_sd, dec_sd = synth.encoder_state_dict(jas, 64, 5), synth.decoder_state_dict(1024, 91, 5)
    eng = QuartzNetCTC(cfg, enc_sd, dec_sd, gemm=gemm)
    r = np.random.RandomState(9)
    lens = np.array([52000, 16000, 33333, 8000, 47999, 160 * 90], dtype=np.int64)
    sig = np.zeros((len(lens), int(lens.max())), dtype=np.float32)
    for b, n in enumerate(lens):
        sig[b, :n] = (0.02 if b % 2 else 0.3) * r.randn(n)        # rows at different levels
    full = eng.forward(torch.from_numpy(sig).to(gpu), torch.from_numpy(lens).to(gpu), want_logp=True, row_independent=True)
    for b in range(len(lens)):
        n = int(lens[b])
        one = eng.forward(torch.from_numpy(sig[b:b + 1, :n].copy()).to(gpu), torch.from_numpy(lens[b:b + 1]).to(gpu),
                          want_logp=True, row_independent=True)
        k = int(one["id_len"][0])
        assert int(full["id_len"][b]) == k and torch.equal(full["ids"][b, :k], one["ids"][0, :k]), b
        f = one["logp"].shape[1]                                   # the frames an unbatched call produces
        assert torch.equal(full["logp"][b, :f], one["logp"][0]), (gemm, b)
    # and in another batch composition (pairs, reversed order)
    rev = eng.forward(torch.from_numpy(sig[::-1].copy()).to(gpu), torch.from_numpy(lens[::-1].copy()).to(gpu),
                      want_logp=True, row_independent=True)
    for b in range(len(lens)):
        f = 1 + int(lens[b]) // 160
        f = (f - 1) // 2 + 1
        assert torch.equal(rev["logp"][len(lens) - 1 - b, :f], full["logp"][b, :f]), (gemm, b)


def test_row_independent_batching_on_random_models_and_batch_shapes(gpu):
    """Forty cases of tests/devtools/fuzz_rows.py: random architecture (or the shipped 12x1), random arithmetic, 2-70 ragged rows at
    different levels (a third of the cases on int16 PCM), every sampled row against its batch-1 call and against the same row in a
    shuffled batch of another size -- ids, id_len and the row's own log-prob frames bit for bit.  Round-6 campaign: 22 832 cases,
    498 406 rows, 0 differences (profiles/r06_fuzz_campaign.txt)."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "devtools"))
    import fuzz_rows
    bad = [m for m in (fuzz_rows.rows_case(c) for c in range(40)) if m]
    assert not bad, bad


def test_beam_final_pass_goes_through_the_lm_score_cache(gpu, tmp_path):
    """pyctcdecode scores the pending word with </s> after the last frame -- unless the text "prefix + word" is already
    in its LM score cache (some earlier frame had ' ' among the candidates while a beam held that prefix and word);
    then the cached score WITHOUT </s> is used.  Two posteriors that end in the same best text, one with and one
    without such a frame: the device must follow the cache in both, and the two scores must differ by the </s> term."""
    from viet_asr_amd.beam import BeamSearchDecoder
    from oracle import beam_oracle as BO
    import test_beam as T
    path, ng = T.toy_lm(str(tmp_path))
    V1 = 29

    def posteriors(space_p):
        # a b _ _ : then a frame where blank dominates and ' ' has probability space_p, then blanks
        rows = []
        for c in (1, 2):
            z = np.full(V1, 1e-6); z[c] = 1.0; rows.append(z)
        z = np.full(V1, 1e-6); z[V1 - 1] = 1.0; z[0] = space_p; rows.append(z)
        for _ in range(3):
            z = np.full(V1, 1e-6); z[V1 - 1] = 1.0; rows.append(z)
        p = np.stack(rows)
        p /= p.sum(1, keepdims=True)
        return np.log(p).astype(np.float32)

    for mode in ("binary", "arpa"):                         # both of pyctcdecode's LM behaviours (test_beam.LM_MODES)
        dec, lm = T.make_decoder(T.LABELS, path, mode), T.oracle_lm(path, mode)
        scores = {}
        for name, sp in (("space_was_a_candidate", 0.02), ("never", 1e-6)):       # token_min_logp = -5 -> p >= 6.7e-3
            lp = posteriors(sp)
            ids, n, score = dec.decode_ids(torch.from_numpy(lp[None]).to(gpu), 16)
            text = dec.decode_batch(torch.from_numpy(lp[None]).to(gpu), 16)[0]
            ref = BO.decode_beams(np.exp(lp.astype(np.float64)), T.LABELS, 16, lm=lm)
            assert text == ref[0][0] == "ab", (name, text, ref[:2])
            assert abs(float(score[0]) - ref[0][2]) < 2e-3, (name, float(score[0]), ref[0][2])
            scores[name] = ref[0][2] - ref[0][1]                    # the LM part of the best hypothesis
        with_eos, _ = lm.score(lm.get_start_state(), "ab", is_last_word=True)
        without, _ = lm.score(lm.get_start_state(), "ab", is_last_word=False)
        assert abs(with_eos - without) > 0.1
        assert abs(scores["never"] - with_eos) < 1e-6 and abs(scores["space_was_a_candidate"] - without) < 1e-6


_FUSED_FUZZ = r"""
import sys, copy, numpy as np, torch
sys.path.insert(0, {tests!r}); sys.path.insert(0, {root!r})
from viet_asr_amd import configs, synth
from viet_asr_amd.engine import QuartzNetCTC
from oracle import quartznet_oracle as O
bad = []
def blk(filters, kernel, repeat, stride=1, residual=False, separable=True):
    return dict(filters=filters, repeat=repeat, kernel=[kernel], stride=[stride], dilation=[1], dropout=0.0,
                residual=residual, separable=separable)
for seed in range({n}):
    rng = np.random.default_rng(4000 + seed)
    cfg = copy.deepcopy(configs.builtin("quartznet15x5"))
    # 256-channel blocks with the two kernel widths the fused kernel covers, repeats 1-5 (a lone sub-block is a residual
    # sub-block), with and without residual, behind a strided or unstrided prologue; a 512-channel block in between
    jas = [blk(256, 33, 1, stride=int(rng.choice([1, 2])))]
    for _ in range(int(rng.integers(1, 4))):
        jas.append(blk(256, int(rng.choice([33, 39])), int(rng.integers(1, 6)), residual=bool(rng.random() < 0.7)))
        if rng.random() < 0.3:
            jas.append(blk(512, 51, 1, residual=True))
            jas.append(blk(256, 39, int(rng.integers(1, 3)), residual=True))
    jas.append(blk(int(rng.choice([128, 256])), 1, 1, separable=False))
    cfg["JasperEncoder"]["jasper"] = jas
    enc_sd = synth.encoder_state_dict(jas, 64, seed)
    dec_sd = synth.decoder_state_dict(jas[-1]["filters"], len(cfg["labels"]) + 1, seed)
    eng = QuartzNetCTC(cfg, enc_sd, dec_sd)
    B, L = int(rng.integers(1, 7)), int(rng.integers(3000, 60000))
    sig, lens = synth.audio_batch(B, L, seed, ragged=True)
    lens[int(rng.integers(0, B))] = L
    lens[int(rng.integers(0, B))] = max(400, int(lens.min()) // 4)      # rows with whole tiles past their length
    for b in range(B):
        sig[b, lens[b]:] = 0
    # per-utterance scales of the bound-based split.  (Not 1e-3: at -80 dBFS the mel energies sink under the log guard and
    # the reference's own (x - mean) / (std + 1e-5) amplifies fp32 rounding noise to 1e-1 in the features -- DESIGN section 2,
    # "conditioning note"; the first version of this test drew that level and failed with the fused kernel switched OFF too.)
    sig[0] *= float(rng.choice([0.03, 1.0, 30.0]))
    ref = O.forward_all(sig, lens, enc_sd, dec_sd, jas)
    eng.handle.profile_begin()
    r = eng.forward(torch.from_numpy(sig).cuda(), torch.from_numpy(lens).cuda(), want_logp=True)
    torch.cuda.synchronize()
    fused = eng.handle.profile_end()["fused"]["launches"]
    want = ref["logp"]
    tol = max(5e-4, 2e-5 * float(want.abs().max()))
    err = float((r["logp"].cpu() - want).abs().max())
    top2 = want.topk(2, -1).values
    clear = (top2[..., 0] - top2[..., 1]) > 2 * tol
    ok = err <= tol and bool((r["pred"].cpu()[clear] == ref["pred"][clear]).all()) and \
        r["enc_len"].cpu().tolist() == ref["enc_len"].tolist() and bool(torch.isfinite(r["logp"]).all())
    want_fused = sum(b["repeat"] for b in jas[1:] if b["separable"] and b["filters"] == 256)
    # (after a 512-channel block the first sub-block has 512 input channels and the residual comes from a 512-channel
    # tensor: neither is a fused shape -- the fuzz found the second case running through the fused kernel with garbage
    # for a K range it does not have; vasr_api.cpp now checks the folded residual's K)
    if not ok or fused == 0:
        bad.append((seed, err, tol, fused, want_fused, B, L))
print("FUSED_FUZZ_OK" if not bad else "FUSED_FUZZ_BAD %r" % bad)
"""


_FUSED_WIDTHS = """
import sys, numpy as np, torch
sys.path.insert(0, {root!r})
import viet_asr_amd
from viet_asr_amd import configs, synth
from viet_asr_amd.engine import QuartzNetCTC
cfg = configs.builtin("quartznet15x5")
jas = cfg["JasperEncoder"]["jasper"]
eng = QuartzNetCTC(cfg, synth.encoder_state_dict(jas, 64, 2), synth.decoder_state_dict(1024, len(cfg["labels"]) + 1, 2))
sig, lens = synth.audio_batch(7, 52000, 2, ragged=True)
eng.handle.profile_begin()
r = eng.forward(torch.from_numpy(sig).cuda(), torch.from_numpy(lens).cuda(), want_logp=True)
torch.cuda.synchronize()
assert eng.handle.profile_end()["fused"]["launches"] == 30
np.save({out!r}, r["logp"].cpu().numpy())
"""


def test_fused_kernel_gives_the_same_bits_on_64_and_128_frame_tiles(gpu, tmp_path):
    """The busy-unit hint of an overlapped beam search (vasr_set_busy_cus) may only choose between forms that give the SAME BITS --
    GEMM tile shapes (test_results_do_not_depend_on_batch_size_or_tile_shape) and the fused kernel's tile WIDTH; whether a sub-block
    is fused at all follows the batch shape alone (round 6: with the hint in that decision, overlapped and serial runs of one batch
    differed).  Here the 15x5 model with every 256-channel sub-block forced through the fused kernel on 64-frame and on 128-frame
    tiles (devtools build): log-probs bit for bit."""
    import subprocess
    import sys
    from viet_asr_amd import _lib
    dev = os.path.join(os.path.dirname(_lib.LIB_PATH), "libvasr_hip_dev.so")
    outs = []
    for tile in ("64", "128"):
        out = str(tmp_path / f"logp{tile}.npy")
        code = _FUSED_WIDTHS.format(root=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), out=out)
        p = subprocess.run([sys.executable, "-c", code], env={**os.environ, "VASR_LIB_PATH": dev, "VASR_FUSED_MIN_TILES": "1", "VASR_FUSED_TILE": tile},
                           capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-2000:])
        outs.append(np.load(out))
    assert outs[0].shape == outs[1].shape and np.array_equal(outs[0], outs[1])


def test_results_hold_next_to_foreign_matrix_kernels_and_across_threads(gpu):
    """Round 6: a second stream running 16-bit matrix kernels -- torch's own fp16 bmm, or this library from another handle -- made
    the front end return wrong values: packed-FP32 vector instructions (the FFT) and FP64 arithmetic (the normalisation statistics)
    of a wavefront are not safe on MI355X while such a kernel shares its compute unit (profiles/r06_concurrency.txt; nothing goes
    through memory).  The STFT kernel is now built without packed-FP32 instructions, the normalisation kernels run alone on their
    compute unit.  Here: the front end, whole forwards in two arithmetics and a beam search, 1.5 s each next to the attacker
    (tests/devtools/stress_attack.py runs every entry point for longer); then two host threads on two handles and two streams
    (stress_threads.py), every result against the idle-device one, bit for bit."""
    import sys
    import threading
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "devtools"))
    import stress_attack
    from viet_asr_amd import configs, stages, synth
    from viet_asr_amd.engine import QuartzNetCTC
    cfg = configs.builtin("quartznet12x1_vi")
    jas = cfg["JasperEncoder"]["jasper"]
    sd = synth.encoder_state_dict(jas, 64, 5), synth.decoder_state_dict(1024, len(cfg["labels"]) + 1, 5)
    eng, eng_b = QuartzNetCTC(cfg, sd[0], sd[1]), QuartzNetCTC(cfg, sd[0], sd[1], gemm="bf16x3")
    sig, lens = synth.audio_batch(8, 48000, 3, ragged=True)
    w, n = torch.from_numpy(sig).to(gpu), torch.from_numpy(lens).to(gpu)
    for name, fn in (("melspec", lambda: stages.melspec(eng.handle, w, n)), ("forward f16x2", lambda: eng.forward(w, n, want_logp=True)),
                     ("forward bf16x3", lambda: eng_b.forward(w, n, want_logp=True))):
        calls, wrong = stress_attack.attack(fn, 1.5)
        assert calls > 50 and wrong == 0, (name, calls, wrong)
    # two threads, two handles, two streams
    pool = []
    for i, (B, L) in enumerate([(1, 30000), (5, 20000), (14, 12000), (40, 9000)]):
        s_, l_ = synth.audio_batch(B, L, 50 + i, ragged=True)
        pool.append((torch.from_numpy(s_).to(gpu), torch.from_numpy(l_).to(gpu)))
    want = [eng.forward(a, b, want_logp=True)["logp"].clone() for a, b in pool]
    torch.cuda.synchronize()
    eng2 = QuartzNetCTC(cfg, sd[0], sd[1])
    stop, wrong, calls = [False], [0, 0], [0, 0]

    def worker(k, e):
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            while not stop[0]:
                for i, (a, b) in enumerate(pool):
                    r = e.forward(a, b, want_logp=True)
                    st.synchronize()
                    calls[k] += 1
                    wrong[k] += int(not torch.equal(r["logp"], want[i]))
    ths = [threading.Thread(target=worker, args=(0, eng)), threading.Thread(target=worker, args=(1, eng2))]
    for t in ths:
        t.start()
    import time
    time.sleep(3.0)
    stop[0] = True
    for t in ths:
        t.join()
    assert min(calls) > 100 and wrong == [0, 0], (calls, wrong)


def test_default_mode_is_run_to_run_deterministic(gpu):
    """The same batch through the default (batched, fused where it pays) path twenty times, a second stream keeping part of the chip
    busy half of the time: log-probs, predictions and ids bit for bit every time (maxima are published as integer max, nothing on the
    greedy path accumulates through float atomics)."""
    from viet_asr_amd import configs, synth
    from viet_asr_amd.engine import QuartzNetCTC
    cfg = configs.builtin("quartznet15x5")
    jas = cfg["JasperEncoder"]["jasper"]
    eng = QuartzNetCTC(cfg, synth.encoder_state_dict(jas, 64, 4), synth.decoder_state_dict(1024, len(cfg["labels"]) + 1, 4))
    sig, lens = synth.audio_batch(40, 40000, 4, ragged=True)
    w, n = torch.from_numpy(sig).to(gpu), torch.from_numpy(lens).to(gpu)
    first = eng.forward(w, n, want_logp=True)
    side = torch.cuda.Stream()
    junk = torch.randn(4096, 4096, device=gpu)
    for i in range(20):
        if i % 2:
            with torch.cuda.stream(side):
                for _ in range(4):
                    junk = (junk @ junk).clamp_(-1, 1)
        r = eng.forward(w, n, want_logp=True)
        assert torch.equal(r["logp"], first["logp"]) and torch.equal(r["pred"], first["pred"]) and torch.equal(r["id_len"], first["id_len"]), i
    torch.cuda.synchronize()


def test_overlapped_beam_search_equals_the_serial_run(gpu, tmp_path):
    """engine.forward_beam(overlap=True): the search of batch k on a side stream under the acoustic pass of batch k + 1, which is
    told the busy compute units.  A sequence of batches of changing size, back to back without a synchronisation, against the same
    batches with overlap=False -- ids, lengths, scores bit for bit (tests/devtools/stress_beam_overlap.py runs this for minutes)."""
    from viet_asr_amd import configs, synth
    from viet_asr_amd.beam import BeamSearchDecoder
    from viet_asr_amd.engine import QuartzNetCTC
    cfg = configs.builtin("quartznet12x1_vi")
    jas = cfg["JasperEncoder"]["jasper"]
    eng = QuartzNetCTC(cfg, synth.encoder_state_dict(jas, 64, 5), synth.decoder_state_dict(1024, len(cfg["labels"]) + 1, 5))
    arpa = str(tmp_path / "lm.arpa")
    synth.synthetic_arpa(arpa, cfg["labels"], n_words=2000, n_bigrams=6000, n_trigrams=6000, seed=1)
    dec = BeamSearchDecoder(cfg["labels"], lm_path=arpa, alpha=0.5, beta=1.5)
    rng = np.random.default_rng(1)
    for _ in range(6):
        seq = []
        for B in (40, 46, 13, 64, 47, 1, 33):
            sig, lens = synth.audio_batch(B, int(rng.integers(9000, 36000)), int(rng.integers(0, 1 << 30)), ragged=True)
            seq.append((torch.from_numpy(sig).to(gpu), torch.from_numpy(lens).to(gpu), int(rng.choice([8, 32, 128]))))
        torch.cuda.synchronize()
        out = [eng.forward_beam(w, n, dec, bw, overlap=True) for w, n, bw in seq]
        torch.cuda.synchronize()
        for (w, n, bw), a in zip(seq, out):
            r = eng.forward_beam(w, n, dec, bw, overlap=False)
            torch.cuda.synchronize()
            assert torch.equal(a["id_len"], r["id_len"]) and torch.equal(a["score"], r["score"]), (w.shape, bw)
            for b in range(w.shape[0]):
                k = int(r["id_len"][b])
                assert torch.equal(a["ids"][b, :k], r["ids"][b, :k]), (w.shape, bw, b)


@pytest.mark.parametrize("tile", ["128", "64"])
def test_fused_depthwise_pointwise_kernel_on_random_shapes(gpu, tile):
    """The fused sub-block kernel -- its 128-frame and (round 4) its 64-frame tile -- forced onto small random workloads
    (devtools build, VASR_FUSED_MIN_TILES=1, VASR_FUSED_TILE): random stacks
    of 256-channel blocks (K = 33 / 39, repeats 1-5, residual or not), ragged batches with rows that leave whole tiles
    empty, utterances at very different levels -- against the oracle with the goldens' tolerance; every case must really
    have gone through the fused kernel (profile class count)."""
    import subprocess
    import sys
    from viet_asr_amd import _lib
    here = os.path.dirname(os.path.abspath(__file__))
    dev = os.path.join(os.path.dirname(_lib.LIB_PATH), "libvasr_hip_dev.so")
    code = _FUSED_FUZZ.format(tests=here, root=os.path.dirname(here), n=10)
    out = subprocess.run([sys.executable, "-c", code], env={**os.environ, "VASR_LIB_PATH": dev, "VASR_FUSED_MIN_TILES": "1", "VASR_FUSED_TILE": tile},
                         capture_output=True, text=True, timeout=900)
    assert "FUSED_FUZZ_OK" in out.stdout, (out.stdout[-3000:], out.stderr[-2000:])


