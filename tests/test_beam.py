"""Beam search: CPU tests of the oracle's own invariants; -m gpu tests compare the device kernel with it.

Parity with the reference is UNPINNED for this path (pyctcdecode + kenlm are third-party and absent); the oracle
restates pyctcdecode's published algorithm and the kernel is held to the oracle: same best transcript, combined
score within 1e-3 (fp64 host vs fp64/fixed-point device)."""
import math
import os

import numpy as np
import pytest
import torch

from oracle import beam_oracle as BO

LABELS = list(" abcdefghijklmnopqrstuvwxyz'")


def toy_lm(tmp_path):
    ng = {("<s>",): (-99.0, -0.3), ("</s>",): (-1.0, 0.0), ("<unk>",): (-2.5, 0.0),
          ("ab",): (-1.2, -0.2), ("ba",): (-1.4, -0.1), ("cab",): (-1.6, -0.2), ("a",): (-1.1, -0.3),
          ("<s>", "ab"): (-0.4, -0.1), ("ab", "ba"): (-0.5, -0.2), ("ba", "</s>"): (-0.3, 0.0), ("ab", "cab"): (-0.7, 0.0),
          ("<s>", "ab", "ba"): (-0.2, 0.0), ("ab", "ba", "</s>"): (-0.1, 0.0)}
    path = os.path.join(tmp_path, "toy.arpa")
    BO.write_arpa(path, 3, ng, no_backoff=("</s>",))        # as KenLM prints it: </s> is in the model, not in the unigram list
    return path, ng


# The LM behaviours of pyctcdecode's build_ctcdecoder (oracle/beam_oracle.py header): "binary" = no unigram list (what the
# reference got from 3-gram-lm.binary), "arpa" = unigram set + character trie read from the ARPA file (what it gets from a
# path ending in .arpa -- the only kind of file this library reads).  Every device test with an LM runs both.
LM_MODES = ["none", "binary", "arpa"]
# the smallest batch the library searches with ONE wavefront per utterance (beam_wave.hip); up to 64 rows an utterance gets four
# wavefronts of a compute unit (beam_group.hip: csrc/beam_group.hip beam_group_width).  Tests reach both forms by batch size.
WAVE_ROWS = 65


def make_decoder(labels, path, mode, alpha=0.7, beta=1.1):
    from viet_asr_amd.beam import BeamSearchDecoder
    return BeamSearchDecoder(labels, lm_path=path if mode != "none" else None, alpha=alpha, beta=beta,
                             unigrams=None if mode == "binary" else "auto")


def oracle_lm(path, mode, alpha=0.7, beta=1.1):
    if mode == "none":
        return None
    return BO.LanguageModel(BO.NgramLM.from_arpa(path), alpha=alpha, beta=beta,
                            unigrams=BO.unigrams_for_path(path) if mode == "arpa" else None)


def random_posteriors(T, V1, seed, peaky=4.0):
    r = np.random.RandomState(seed)
    z = r.randn(T, V1) * peaky
    z[:, -1] += 2.0                                  # blank-heavy like a CTC model
    z[:, 0] += 1.0                                   # spaces occur
    lp = z - np.log(np.exp(z).sum(1, keepdims=True))
    return lp.astype(np.float32)


def test_oracle_without_lm_reduces_to_greedy_on_peaky_input():
    lp = np.full((6, 29), -30.0, dtype=np.float32)
    for t, c in enumerate([1, 1, 28, 2, 0, 1]):       # a a _ b ' ' a
        lp[t, c] = 0.0
    lp = lp - np.log(np.exp(lp.astype(np.float64)).sum(1, keepdims=True)).astype(np.float32)
    assert BO.decode(lp, LABELS, 8) == "ab a"


def test_oracle_merges_prefixes_by_logsumexp():
    # two frames, classes {a, blank}: P("a") = 1 - P(blank,blank)
    p = np.array([[0.0, 0.6, 0.4], [0.0, 0.3, 0.7]])           # [' ', 'a', blank]
    beams = BO.decode_beams(p, [" ", "a"], 8, token_min_logp=-50)
    got = {t: s for t, s, _ in beams}
    assert abs(math.exp(got["a"]) - (1 - 0.4 * 0.7)) < 1e-9 and abs(math.exp(got[""]) - 0.28) < 1e-9


def test_arpa_round_trip_and_backoff(tmp_path):
    path, ng = toy_lm(str(tmp_path))
    lm = BO.NgramLM.from_arpa(path)
    assert lm.order == 3 and len(lm.ngrams) == len(ng)
    s, st = lm.base_score(("<s>",), "ab")
    assert abs(s - (-0.4)) < 1e-6 and st == ("<s>", "ab")
    s, _ = lm.base_score(("<s>", "ab"), "cab")                  # trigram missing -> bo(<s> ab) + p(cab|ab)
    assert abs(s - (-0.1 + -0.7)) < 1e-6
    s, _ = lm.base_score(("ab", "cab"), "ba")                   # back off twice to the unigram
    assert abs(s - (0.0 + -0.2 + -1.4)) < 1e-6
    s, _ = lm.base_score(("<s>",), "zzz")                       # OOV -> <unk>
    assert abs(s - (-0.3 + -2.5)) < 1e-6
    from viet_asr_amd.beam import read_arpa
    order, ng2 = read_arpa(path)
    assert order == 3 and set(ng2) == set(ng)
    with open(os.path.join(str(tmp_path), "x.binary"), "wb") as f:
        f.write(b"mmap lm http://kheafield.com/code format version 5\n\0")
    with pytest.raises(NotImplementedError):
        read_arpa(os.path.join(str(tmp_path), "x.binary"))
    # the reference's default lm_path IS such a file (infer.py:184): a decoder built on it must not lose the LM silently
    from viet_asr_amd.beam import BeamSearchDecoder
    with pytest.raises(ValueError, match="ARPA"):
        BeamSearchDecoder(LABELS, lm_path=os.path.join(str(tmp_path), "x.binary"))
    with pytest.raises(ValueError, match="lmplz"):
        BeamSearchDecoder(LABELS, lm_path=os.path.join(str(tmp_path), "no_such_file.arpa"))
    with pytest.warns(UserWarning, match="without a language model"):
        d = BeamSearchDecoder(LABELS, lm_path=os.path.join(str(tmp_path), "x.binary"), allow_missing_lm=True)
    assert d.lm_path is None
    assert BeamSearchDecoder(LABELS, lm_path=path).lm_path == path
    # the reference's shipped call site (infer.py:184 passes a `.binary`): with the ARPA source NEXT to it under the same stem the
    # unmodified path works, and keeps the behaviour the binary has in pyctcdecode (no unigram list)
    import shutil
    shutil.copy(path, os.path.join(str(tmp_path), "x.arpa"))
    with pytest.warns(UserWarning, match="reading its ARPA source"):
        d = BeamSearchDecoder(LABELS, lm_path=os.path.join(str(tmp_path), "x.binary"))
    assert d.lm_path == os.path.join(str(tmp_path), "x.arpa") and d.unigrams is None
    assert BeamSearchDecoder(LABELS, lm_path=os.path.join(str(tmp_path), "x.arpa")).unigrams == "auto"


def test_lm_changes_the_ranking(tmp_path):
    path, _ = toy_lm(str(tmp_path))
    lm = BO.LanguageModel(BO.NgramLM.from_arpa(path), alpha=2.0, beta=0.5)
    lp = random_posteriors(30, 29, 5, peaky=2.0)
    a = BO.decode_beams(np.exp(lp.astype(np.float64)), LABELS, 16)
    b = BO.decode_beams(np.exp(lp.astype(np.float64)), LABELS, 16, lm=lm)
    assert a[0][2] == a[0][1] and b[0][2] != b[0][1]              # combined score carries the LM term


def mode_lm(tmp_path):
    """A bigram model over {ab, ba, a}: "ab" and "a" carry a back-off weight (=> in pyctcdecode's unigram list for an .arpa
    path), "ba" does not (=> in the model, outside the list)."""
    ng = {("<s>",): (-99.0, -0.3), ("</s>",): (-1.0, 0.0), ("<unk>",): (-2.5, 0.0),
          ("ab",): (-1.2, -0.2), ("ba",): (-1.2, 0.0), ("a",): (-1.5, -0.1),
          ("<s>", "ab"): (-0.6, 0.0), ("<s>", "ba"): (-0.6, 0.0)}
    path = os.path.join(tmp_path, "mode.arpa")
    BO.write_arpa(path, 2, ng, no_backoff=("</s>", "ba"))
    return path


MODE_LABELS = [" ", "a", "b"]


def mode_case_set_membership():
    """"ba" is acoustically ahead of "ab" (0.55^2 vs 0.45^2); both are words of the model with the same probabilities."""
    p = np.array([[0.0, 0.45, 0.55, 0.0], [0.0, 0.0, 0.0, 1.0], [0.0, 0.55, 0.45, 0.0]]) + 1e-9
    return np.log(p / p.sum(1, keepdims=True)).astype(np.float32)


def mode_case_partial_word_prune():
    """Frame 0 prefers blank (0.61) to 'a' (0.38), frame 1 is 'b': the beam "a" needs to survive frame 0 for "ab" to exist."""
    p = np.array([[0.005, 0.38, 0.005, 0.61], [0.0067, 0.0067, 0.98, 0.0066]])
    return np.log(p / p.sum(1, keepdims=True)).astype(np.float32)


def test_unigram_list_is_read_as_pyctcdecode_reads_it(tmp_path):
    from viet_asr_amd import beam
    path = mode_lm(str(tmp_path))
    want = {"<s>", "<unk>", "ab", "a"}                       # three tab-separated fields; "ba" and "</s>" have two
    assert BO.load_unigram_set_from_arpa(path) == want == beam.load_unigram_set_from_arpa(path)
    assert BO.unigrams_for_path(path) == want and BO.unigrams_for_path(path[:-5] + ".binary") is None
    lm = BO.LanguageModel(BO.NgramLM.from_arpa(path), unigrams=want)
    assert lm._unigram_set == {"<s>", "ab", "a"}             # "t in kenlm_model": <unk> is vocabulary index 0
    assert lm._prefixes == {"<", "<s", "<s>", "a", "ab"}
    assert lm.score_partial_token("a") == 0.0 and lm.score_partial_token("b") == -10.0
    assert lm.score_partial_token("abababab") == -10.0 * 8 / 6
    none = BO.LanguageModel(BO.NgramLM.from_arpa(path))
    assert none.score_partial_token("a") == -10.0
    # a unigram-only ARPA prints no back-off at all: pyctcdecode finds no unigrams and raises
    one = os.path.join(str(tmp_path), "one.arpa")
    BO.write_arpa(one, 1, {("<s>",): (-99.0, 0.0), ("</s>",): (-1.0, 0.0), ("<unk>",): (-2.0, 0.0), ("a",): (-1.0, 0.0)})
    for f in (BO.load_unigram_set_from_arpa, beam.load_unigram_set_from_arpa):
        with pytest.raises(ValueError):
            f(one)


def test_the_two_lm_behaviours_rank_differently(tmp_path):
    """pyctcdecode's build_ctcdecoder on an .arpa path (unigram set + character trie) against the same model as a .binary
    (no unigram list), on inputs where the difference is provable by hand."""
    path = mode_lm(str(tmp_path))
    # (1) a committed word outside the unigram list gets the unk offset only in arpa mode: alpha * -10 * ln 10 = -11.5
    lp = mode_case_set_membership()
    got = {m: BO.decode_beams(np.exp(lp.astype(np.float64)), MODE_LABELS, 8, lm=oracle_lm(path, m, 0.5, 1.5)) for m in ("binary", "arpa")}
    assert got["binary"][0][0] == "ba" and got["arpa"][0][0] == "ab"
    wide = {m: BO.decode_beams(np.exp(lp.astype(np.float64)), MODE_LABELS, 8, lm=oracle_lm(path, m, 0.5, 1.5), beam_prune_logp=-50.0)
            for m in ("binary", "arpa")}
    s_ba = {m: [b[2] for b in wide[m] if b[0] == "ba"][0] for m in wide}
    assert abs((s_ba["binary"] - s_ba["arpa"]) - 0.5 * 10.0 * math.log(10.0)) < 1e-9
    # (2) a partial word that is a prefix of a known word carries no penalty in arpa mode; in binary mode its -10 meets
    #     beam_prune_logp = -10 and the beam is gone before the word can be completed
    lp = mode_case_partial_word_prune()
    got = {m: BO.decode_beams(np.exp(lp.astype(np.float64)), MODE_LABELS, 8, lm=oracle_lm(path, m, 0.5, 1.5)) for m in ("binary", "arpa")}
    assert got["binary"][0][0] == "b" and "ab" not in [b[0] for b in got["binary"]]
    assert got["arpa"][0][0] == "ab"


def test_trie_buckets_hold_every_node_where_the_kernel_looks(tmp_path):
    """viet_asr_amd.beam._trie_buckets against csrc/beam_common.h trie_has_node: from the home bucket, bucket by bucket, until
    the key or a bucket with a free cell."""
    from viet_asr_amd import beam
    r = np.random.RandomState(3)
    for n in (0, 1, 7, 500, 40000):
        keys = np.unique(r.randint(0, 2**63, size=n, dtype=np.int64).astype(np.uint64) | np.uint64(1))
        cells = beam._trie_buckets(keys)
        nb = len(cells)
        assert nb >= 16 and nb & (nb - 1) == 0 and 2 * nb >= 4 * len(keys)
        assert sorted(cells[cells != 0].tolist()) == sorted(keys.tolist())
        def has(k):
            i = int(beam._home(np.array([k], dtype=np.uint64), nb)[0])
            for _ in range(nb):
                if k in cells[i]:
                    return True
                if (cells[i] == 0).any():
                    return False
                i = (i + 1) & (nb - 1)
            return False
        for k in keys[:300]:
            assert has(k)
        for k in (r.randint(0, 2**63, size=100, dtype=np.int64).astype(np.uint64) | np.uint64(1)):
            assert has(k) == (k in keys)
        full = (cells != 0).all(1).mean() if nb else 0.0
        assert full <= 0.12, full                              # a full bucket is what costs a lookup a second load


# ------------------------------------------------------------------------------------------- device
def test_lm_tables_spread_their_keys_and_probe_chains_stay_short():
    """The device n-gram tables (include/vasr.h vasr_lm_create, csrc/beam_common.h lm_home): power-of-two capacity, home slot
    by Fibonacci hashing of the xor-folded key.  The keys are FNV-style products of SMALL word ids -- their entropy sits in
    bits 0-24 and 40+ -- so any plain bit field clusters (round 4's first layout used bits 17.. and sent 20 000 unigrams to
    256 home slots: every lookup walked thousands of entries, the search took seconds).  Pinned here: the builder's home
    slots spread and no key sits more than a few dozen slots from home, for unigram-, bigram- and trigram-shaped keys."""
    from viet_asr_amd import beam
    h0 = np.uint64(1469598103934665603)
    r = np.random.RandomState(0)
    for n_words, order, count in ((20000, 1, 20000), (20000, 2, 50000), (20000, 3, 50000), (300, 2, 5000)):
        ids = np.stack([r.randint(0, n_words, count) for _ in range(order)], 1).astype(np.uint64)
        ids = np.unique(ids, axis=0)
        h = np.full(len(ids), h0, dtype=np.uint64)
        for i in reversed(range(order)):
            h = beam._hstep_np(h, ids[:, i])
        keys = h | np.uint64(1)
        cap = beam._cap(len(keys))
        assert cap & (cap - 1) == 0 and cap >= 2 * len(keys)
        home = beam._home(keys, cap)
        assert home.min() >= 0 and home.max() < cap
        assert np.bincount(home, minlength=cap).max() <= 8, (n_words, order)
        slots, where = beam._table(keys, cap)
        assert (slots[where] == keys).all()
        assert int(((where - home) % cap).max()) <= 48, (n_words, order)
        # a lookup as the kernel does it: from home, linearly, until the key or an empty slot
        for k, w, hm in list(zip(keys, where, home))[:200]:
            i = int(hm)
            while slots[i] != k:
                assert slots[i] != 0
                i = (i + 1) & (cap - 1)
            assert i == w


def test_fixed_point_term_integer_form():
    """csrc/beam_group.hip fix44: the 2^-44 fixed-point term of a prefix merge, (u64)((double)e * 2^44) for a float e in
    [0, 1], as integer arithmetic on the float's bits (mantissa shifted by E - 106).  Exhaustive agreement over every float
    in [0, 1] was checked once on the host in C; this replays both forms in numpy on 4e6 random floats, the powers of two,
    their neighbours, 0 and 1."""
    r = np.random.default_rng(0)
    bits = np.concatenate([r.integers(0x00800000, 0x3f800001, 4_000_000, dtype=np.uint32),
                           np.array([0, 0x3f800000, 0x3f7fffff, 0x00800000], dtype=np.uint32),
                           (np.arange(1, 128, dtype=np.uint32) << 23), (np.arange(1, 128, dtype=np.uint32) << 23) - 1,
                           (np.arange(1, 127, dtype=np.uint32) << 23) + 1]).astype(np.uint32)
    bits = bits[(bits <= 0x3f800000) & (((bits >> 23) & 0xff) > 0) | (bits == 0)]
    e = bits.view(np.float32)
    ref = np.floor(e.astype(np.float64) * 2.0 ** 44).astype(np.uint64)
    E = ((bits >> 23) & 0xff).astype(np.int64)
    m = ((bits & 0x7fffff) | 0x800000).astype(np.uint64)
    sh = E - 106
    left = m << np.clip(sh, 0, 63).astype(np.uint64)
    right = m >> np.clip(-sh, 0, 63).astype(np.uint64)
    got = np.where(E == 0, 0, np.where(sh >= 0, left, right)).astype(np.uint64)
    assert np.array_equal(got, ref)


def test_reciprocal_forms_of_the_wave_kernels_integer_divisions():
    """csrc/beam_wave.hip divides by a wave-uniform small integer through the hardware reciprocal (1 ulp): pair index ->
    (beam, candidate) as (int)((p + 0.5f) * rcp(nc)) and candidates per pass as (int)((kFill + 0.5f) * rcp(nb)).  Replayed
    in float32 with the reciprocal perturbed by +-2 ulp: the quotient never moves."""
    kfill = 512 * 7 // 10
    for d in range(1, 129):
        inv = np.float32(1.0) / np.float32(d)
        for ulp in (-2, -1, 0, 1, 2):
            rcp = np.nextafter(inv, np.float32(np.inf if ulp > 0 else -np.inf)) if abs(ulp) == 1 else inv
            if abs(ulp) == 2:
                rcp = np.nextafter(np.nextafter(inv, np.float32(np.inf if ulp > 0 else -np.inf)), np.float32(np.inf if ulp > 0 else -np.inf))
            p = np.arange(0, 1024, dtype=np.float32)
            assert (((p + np.float32(0.5)) * rcp).astype(np.int32) == np.arange(1024) // d).all(), d
            assert int((np.float32(kfill) + np.float32(0.5)) * rcp) == kfill // d, d


@pytest.mark.gpu
@pytest.mark.parametrize("lm_mode", LM_MODES)
@pytest.mark.parametrize("beam_width,V1,seed", [(8, 29, 1), (32, 29, 2), (128, 29, 3), (20, 91, 4)])
def test_device_beam_search_matches_oracle(gpu, tmp_path, lm_mode, beam_width, V1, seed):
    from viet_asr_amd.beam import BeamSearchDecoder
    labels = LABELS if V1 == 29 else [" "] + [chr(0x100 + i) for i in range(89)]
    path, _ = toy_lm(str(tmp_path))
    lp = np.stack([random_posteriors(40 + 7 * b, V1, seed * 10 + b)[:40] for b in range(3)])
    dec = make_decoder(labels, path, lm_mode)
    ids, n, score = dec.decode_ids(torch.from_numpy(lp).to(gpu), beam_width)
    texts = dec.decode_batch(torch.from_numpy(lp).to(gpu), beam_width)
    lm = oracle_lm(path, lm_mode)
    for b in range(3):
        ref = BO.decode_beams(np.exp(lp[b].astype(np.float64)), labels, beam_width, lm=lm)
        # near-ties between the two best hypotheses may legitimately resolve differently (fp rounding)
        close = len(ref) > 1 and abs(ref[0][2] - ref[1][2]) < 1e-3
        assert texts[b] == ref[0][0] or (close and texts[b] == ref[1][0]), (texts[b], ref[:2])
        if texts[b] == ref[0][0]:
            assert abs(float(score[b]) - ref[0][2]) < 2e-3 * max(1.0, abs(ref[0][2]) / 50), (float(score[b]), ref[0][2])


@pytest.mark.gpu
@pytest.mark.parametrize("rows", [1, WAVE_ROWS])
def test_device_beam_search_has_both_lm_behaviours(gpu, tmp_path, rows):
    """The hand-checkable inputs of test_the_two_lm_behaviours_rank_differently on the device, in both kernel forms (1 row:
    four wavefronts per utterance, 65 rows: one), and the default (`unigrams="auto"`) follows the path's suffix as
    pyctcdecode's build_ctcdecoder does (beam_search_decoder.py:82-87: "either .arpa or .bin file")."""
    import shutil
    from viet_asr_amd.beam import BeamSearchDecoder
    path = mode_lm(str(tmp_path))
    other = os.path.join(str(tmp_path), "same_model.txt")       # ARPA text under a suffix pyctcdecode reads no unigrams from
    shutil.copy(path, other)
    for lp, want in ((mode_case_set_membership(), {"binary": "ba", "arpa": "ab"}),
                     (mode_case_partial_word_prune(), {"binary": "b", "arpa": "ab"})):
        x = torch.from_numpy(np.repeat(lp[None], rows, 0)).to(gpu)
        for mode in ("binary", "arpa"):
            dec = make_decoder(MODE_LABELS, path, mode, 0.5, 1.5)
            texts = dec.decode_batch(x, 8)
            ref = BO.decode_beams(np.exp(lp.astype(np.float64)), MODE_LABELS, 8, lm=oracle_lm(path, mode, 0.5, 1.5))
            assert texts == [want[mode]] * rows == [ref[0][0]] * rows, (mode, texts, ref[:2])
            assert abs(float(dec.decode_ids(x, 8)[2][0]) - ref[0][2]) < 2e-3
        assert BeamSearchDecoder(MODE_LABELS, lm_path=path, alpha=0.5, beta=1.5).decode_batch(x, 8) == [want["arpa"]] * rows
        assert BeamSearchDecoder(MODE_LABELS, lm_path=other, alpha=0.5, beta=1.5).decode_batch(x, 8) == [want["binary"]] * rows


@pytest.mark.gpu
def test_beam_module_on_model_output(gpu):
    """BeamSearchDecoderWithLM NeuralModule on real log-probs of the synthetic model; beam 1 without LM and a
    large prune window must reproduce the best path when posteriors are peaky."""
    from conftest import load_golden
    from viet_asr_amd import asr as nemo_asr
    from viet_asr_amd.core import DeviceType, NeuralModuleFactory
    g, cfg, *_ = load_golden("en15x5_b2_ragged")
    NeuralModuleFactory(placement=DeviceType.GPU)
    beam = nemo_asr.BeamSearchDecoderWithLM(vocab=cfg["labels"], beam_width=16, alpha=0.5, beta=1.5, lm_path=None, num_cpus=1)
    logp = torch.from_numpy(g["logp"]).to(gpu)
    out = beam(force_pt=True, log_probs=logp, log_probs_length=None)
    assert isinstance(out, list) and len(out) == 2
    for b in range(2):
        assert out[b] == BO.decode(g["logp"][b], cfg["labels"], 16)
    single = beam(force_pt=True, log_probs=logp[:1], log_probs_length=None)
    assert isinstance(single, str) and single == out[0]            # what infer.py consumes: evaluated_tensors[0][0]


@pytest.mark.gpu
@pytest.mark.parametrize("lm_mode", LM_MODES)
def test_device_beam_search_flat_posteriors_long(gpu, tmp_path, lm_mode):
    """Flat posteriors over 150 frames: every class clears token_min_logp, 64 beams x 29 candidates = 1856 pairs do not
    fit the merge table (1434), so every frame takes two candidate passes with the first pass's survivors carried into
    the second selection; multi-digit radix select and heavy prefix merging are all on the path.  Compared with the
    UNMODIFIED restatement of pyctcdecode (round 2 compared with an oracle capped like the kernel was)."""
    from viet_asr_amd.beam import BeamSearchDecoder
    path, _ = toy_lm(str(tmp_path))
    lp = np.stack([random_posteriors(150, 29, 70 + b, peaky=1.0) for b in range(2)])
    dec = make_decoder(LABELS, path, lm_mode)
    ids, n, score = dec.decode_ids(torch.from_numpy(lp).to(gpu), 64)
    texts = dec.decode_batch(torch.from_numpy(lp).to(gpu), 64)
    lm = oracle_lm(path, lm_mode)
    for b in range(2):
        ref = BO.decode_beams(np.exp(lp[b].astype(np.float64)), LABELS, 64, lm=lm)
        close = len(ref) > 1 and abs(ref[0][2] - ref[1][2]) < 1e-3
        assert texts[b] == ref[0][0] or (close and texts[b] == ref[1][0]), (texts[b], ref[:2])
        if texts[b] == ref[0][0]:
            assert abs(float(score[b]) - ref[0][2]) < 2e-3 * max(1.0, abs(ref[0][2]) / 50), (float(score[b]), ref[0][2])


def ctc_like_posteriors(T, V1, seed, p_blank=0.7):
    """Long runs of frames where only blank clears token_min_logp, separated by frames with one dominant character and
    a few alternatives: what a converged CTC model emits."""
    r = np.random.RandomState(seed)
    z = r.randn(T, V1)
    blank = r.rand(T) < p_blank
    z[blank, -1] += 14.0
    idx = np.where(~blank)[0]
    z[idx, r.randint(0, V1 - 1, len(idx))] += 6.0
    z[idx[::5], 0] += 6.0                            # word boundaries, so that the LM is consulted
    lp = z - np.log(np.exp(z).sum(1, keepdims=True))
    return lp.astype(np.float32)


@pytest.mark.gpu
@pytest.mark.parametrize("lm_mode", LM_MODES)
@pytest.mark.parametrize("beam_width", [4, 32, 128])
def test_device_beam_search_blank_runs(gpu, tmp_path, lm_mode, beam_width):
    """Blank-only frames on beams that all end in blank take an early exit in the kernel (scores shift, nothing else
    changes); the first frame of every run must still merge beams that differ in their last character only."""
    from viet_asr_amd.beam import BeamSearchDecoder
    path, _ = toy_lm(str(tmp_path))
    lp = np.stack([ctc_like_posteriors(120, 29, 300 + b, p_blank=(0.5, 0.7, 0.9, 1.0)[b]) for b in range(4)])
    dec = make_decoder(LABELS, path, lm_mode)
    ids, n, score = dec.decode_ids(torch.from_numpy(lp).to(gpu), beam_width)
    texts = dec.decode_batch(torch.from_numpy(lp).to(gpu), beam_width)
    lm = oracle_lm(path, lm_mode)
    for b in range(4):
        ref = BO.decode_beams(np.exp(lp[b].astype(np.float64)), LABELS, beam_width, lm=lm)
        close = len(ref) > 1 and abs(ref[0][2] - ref[1][2]) < 1e-3
        assert texts[b] == ref[0][0] or (close and texts[b] == ref[1][0]), (b, texts[b], ref[:2])
        if texts[b] == ref[0][0]:
            assert abs(float(score[b]) - ref[0][2]) < 2e-3 * max(1.0, abs(ref[0][2]) / 50), (float(score[b]), ref[0][2])


@pytest.mark.gpu
@pytest.mark.parametrize("lm_mode", LM_MODES)
def test_device_beam_search_per_row_frame_counts(gpu, tmp_path, lm_mode):
    """vasr_beam_search_rows_f32: a row searched over its own frame count inside a padded batch gives exactly what the
    truncated row gives alone (ids and score bit-identical; 0 frames -> empty hypothesis)."""
    from viet_asr_amd.beam import BeamSearchDecoder
    path, _ = toy_lm(str(tmp_path))
    T = 90
    lp = np.stack([ctc_like_posteriors(T, 29, 500 + b, p_blank=0.6) for b in range(5)])
    frames = [T, 37, 1, 64, 0]
    dec = make_decoder(LABELS, path, lm_mode)
    x = torch.from_numpy(lp).to(gpu)
    ids, n, score = dec.decode_ids(x, 32, frames=frames)
    assert int(n[4]) == 0
    for b in range(4):
        ids1, n1, score1 = dec.decode_ids(x[b : b + 1, : frames[b]].contiguous(), 32)
        assert int(n[b]) == int(n1[0]) and torch.equal(ids[b, : n[b]], ids1[0, : n1[0]])
        assert float(score[b]) == float(score1[0])
    assert dec.decode_batch(x, 32, frames=frames)[1] == dec.decode_batch(x[1:2, :37].contiguous(), 32)[0]
    with pytest.raises(ValueError):
        dec.decode_ids(x, 32, frames=[1, 2])


@pytest.mark.gpu
def test_beam_search_writes_stay_inside_its_buffers(gpu):
    """Workspace (vasr_beam_workspace_bytes) and the [B][T] / [B] outputs between sentinel-filled guard regions."""
    from viet_asr_amd import _lib
    B, T, V1, G = 3, 77, 29, 1 << 16
    lp = torch.from_numpy(np.stack([random_posteriors(T, V1, 900 + b, peaky=1.5) for b in range(B)])).to(gpu)
    L = _lib.lib()
    sizes = dict(ws=int(L.vasr_beam_workspace_bytes(B, T)), ids=B * T * 4, n=B * 4, score=B * 4)
    bufs = {k: torch.full((((n + 15) // 16) * 16 + 2 * G,), 0xA5, dtype=torch.uint8, device=gpu) for k, n in sizes.items()}
    p = {k: v[G:].data_ptr() for k, v in bufs.items()}
    _lib.check(L.vasr_beam_search_f32(lp.data_ptr(), B, T, V1, 0, 128, -5.0, -10.0, None, p["ids"], p["n"], p["score"],
                                      p["ws"], sizes["ws"], torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    for k, buf in bufs.items():
        assert bool((buf[:G] == 0xA5).all()) and bool((buf[G + sizes[k]:] == 0xA5).all()), f"write outside {k}"
    assert int(bufs["n"][G : G + 4 * B].view(torch.int32).min()) > 0


@pytest.mark.gpu
def test_device_beam_search_randomised_cases(gpu):
    """Four hundred cases of tests/devtools/fuzz_beam.py (posterior shape, length, beam width, LM and its weights all drawn
    at random) against the oracle -- the whole fuzz run is in the driver's suite since round 4 (sixty cases before; the
    one-wavefront kernel takes 3 s for all of them)."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "devtools"))
    import fuzz_beam
    bad = [m for m in (fuzz_beam.run_case(c) for c in range(400)) if m]
    assert not bad, bad
    from test_gpu_parity import _record
    _record("beam_fuzz", **fuzz_beam.STATS)      # how many of the cases the < 1e-3 near-tie allowance decided
    assert fuzz_beam.STATS["near_tie_excuses"] == 0     # measured (round 5): none of 2 400 cases; the allowance stays for other hosts' numpy
    # round 6: a 34 738-case campaign met ONE disagreement -- a beam cut whose kept and dropped beams were 4.2e-6 apart at score
    # -193.57; the oracle re-run with that one cut broken the other way returned the device's text and score to the digit
    # (fuzz_beam.cut_near_tie).  None of these 400 cases needs that analysis:
    assert fuzz_beam.STATS["cut_near_tie_excuses"] == 0


@pytest.mark.gpu
def test_beam_module_through_the_executor_with_a_batch(gpu):
    """The reference's wiring (infer.py:146-160) through NeuralModuleFactory.infer with B = 3 ragged utterances: the
    beam module's single output port carries ONE value per batch -- a list of B transcripts (round 1 zipped that list
    against the port and kept row 0 only) -- and every row is searched over its own encoded length, not over the
    padding frames of the batch (log_probs_length, which the reference ignores because it only ever runs B = 1)."""
    from viet_asr_amd import asr as nemo_asr
    from viet_asr_amd import configs, synth
    from viet_asr_amd.beam import BeamSearchDecoder
    from viet_asr_amd.core import DeviceType, NeuralModuleFactory
    cfg = configs.builtin("quartznet12x1_vi")
    jas = cfg["JasperEncoder"]["jasper"]
    enc_sd, dec_sd = synth.encoder_state_dict(jas, 64, 12), synth.decoder_state_dict(1024, 91, 12)
    nf = NeuralModuleFactory(placement=DeviceType.GPU)
    dl = nemo_asr.AudioDataLayer(sample_rate=16000)
    pre = nemo_asr.AudioToMelSpectrogramPreprocessor(**dict(cfg["AudioToMelSpectrogramPreprocessor"], dither=0, pad_to=0))
    enc = nemo_asr.JasperEncoder(feat_in=64, **cfg["JasperEncoder"])
    dec = nemo_asr.JasperDecoderForCTC(feat_in=1024, num_classes=len(cfg["labels"]))
    beam = nemo_asr.BeamSearchDecoderWithLM(vocab=cfg["labels"], beam_width=16, alpha=0.5, beta=1.5, lm_path=None, num_cpus=1)
    enc.load_state_dict({k: torch.as_tensor(v) for k, v in enc_sd.items()})
    dec.load_state_dict({k: torch.as_tensor(v) for k, v in dec_sd.items()})
    a, al = dl()
    mel, ml = pre(input_signal=a, length=al)
    e, el = enc(audio_signal=mel, length=ml)
    lp = dec(encoder_output=e)
    hyp = beam(log_probs=lp, log_probs_length=el)
    sig, lens = synth.audio_batch(3, 31990, 12, ragged=True)      # not a multiple of the hop: T' == enc_len of the longest row
    lens[:] = [31990, 9000, 20000]
    for b in range(3):
        sig[b, lens[b]:] = 0
    dl.set_batch([sig[b, : lens[b]] for b in range(3)])
    lp_v, el_v, hyp_v = [o[0] for o in nf.infer(tensors=[lp, el, hyp], verbose=False)]
    assert isinstance(hyp_v, list) and len(hyp_v) == 3 and all(isinstance(t, str) for t in hyp_v)
    direct = BeamSearchDecoder(cfg["labels"])
    assert hyp_v == direct.decode_batch(lp_v.to(gpu), 16, frames=[int(v) for v in el_v.tolist()])
    for b in range(3):                         # row b over its own frames == a batch-1 search of exactly those frames
        assert hyp_v[b] == direct.decode_batch(lp_v[b:b + 1, : int(el_v[b])].to(gpu), 16)[0]
    whole = direct.decode_batch(lp_v.to(gpu), 16)                    # all padded frames: what round 1 searched
    assert whole[0] == hyp_v[0]                                      # the longest row has no padding
    dl.set_signal(sig[1, : lens[1]])
    single = nf.infer(tensors=[hyp], verbose=False)[0][0]
    assert isinstance(single, str)                                   # what infer.py consumes: evaluated_tensors[0][0]


_AB_SNIPPET = r"""
import os, sys
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import numpy as np, torch
import viet_asr_amd
from viet_asr_amd.beam import BeamSearchDecoder
import test_beam as T
path, _ = T.toy_lm({tmp!r})
out = []
for V1, labels in ((29, T.LABELS),):
    lp = np.stack([T.ctc_like_posteriors(160, V1, 900 + b, p_blank=0.55) for b in range(6)] +
                  [T.random_posteriors(160, V1, 950 + b, peaky=k) for b, k in enumerate((0.5, 2.0, 4.0))])
    x = torch.from_numpy(lp).cuda()
    for mode in T.LM_MODES:
        dec = T.make_decoder(labels, path, mode)
        for w in (8, 50, 100, 128):
            ids, n, score = dec.decode_ids(x, w)
            out.append((ids.cpu().numpy(), n.cpu().numpy(), score.cpu().numpy()))
np.save({dst!r}, np.array([(a.tobytes(), b.tobytes(), c.tobytes()) for a, b, c in out], dtype=object), allow_pickle=True)
"""


@pytest.mark.gpu
def test_wave_kernel_and_group_kernel_agree(gpu, tmp_path):
    """beam_wave.hip (one wavefront per utterance: batches) against beam_group.hip (an utterance on four wavefronts of a
    compute unit: the serving latency; VASR_BEAM_GROUP pins the form in the devtools build): the same keys, merge arithmetic
    (ordered-int max, fixed-point sums), prune, radix select and rank rules on a different schedule and with twice the pairs
    per pass -- hypotheses, lengths and scores identical bit for bit on CTC-like, flat and peaked posteriors, widths 8 ... 128,
    without an LM and with one in both of pyctcdecode's behaviours (flat posteriors at width 100-128 take several passes per frame and a radix select on most).
    Each form runs in its own process (the switch is read once)."""
    import subprocess, sys
    from conftest import ROOT
    dev = os.path.join(ROOT, "viet-asr_amd", "lib", "libvasr_hip_dev.so")
    res = []
    for tag, extra in (("wave", {"VASR_BEAM_GROUP": "0"}), ("group4", {"VASR_BEAM_GROUP": "4"})):
        dst = str(tmp_path / f"{tag}.npy")
        env = dict(os.environ, VASR_LIB_PATH=dev, **extra)
        r = subprocess.run([sys.executable, "-c", _AB_SNIPPET.format(root=ROOT, tmp=str(tmp_path), dst=dst)], env=env,
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        res.append(np.load(dst, allow_pickle=True))
    assert len(res[0]) == len(res[1]) == 12
    for other in (1,):
        for k, (a, b) in enumerate(zip(res[0], res[other])):
            n_a, n_b = np.frombuffer(a[1], np.int32), np.frombuffer(b[1], np.int32)
            assert (n_a == n_b).all(), (other, k)
            ids_a = np.frombuffer(a[0], np.int32).reshape(len(n_a), -1)
            ids_b = np.frombuffer(b[0], np.int32).reshape(len(n_b), -1)
            for r_ in range(len(n_a)):
                assert (ids_a[r_, : n_a[r_]] == ids_b[r_, : n_b[r_]]).all(), (other, k, r_)
            assert a[2] == b[2], (other, k)            # scores, bit for bit


@pytest.mark.gpu
def test_exact_score_ties_do_not_depend_on_the_kernel_form(gpu, tmp_path):
    """Almost flat posteriors (log-probs within 0.06 of each other, up to 128 classes) make hypotheses of DIFFERENT texts reach
    the same 64-bit score, also at the beam cut.  The two kernels do not hold their entries in the same order (passes of 358
    and of 716 pairs; the claimer of a merged prefix), so a tie broken by position gave different survivors: tools/soak_beam.py
    found 3 such cases in 2 000 = 32 000 utterances (seeds below; every other posterior shape: none).  Since then a tie at the cut, and a tie
    of the final scores, goes to the larger table key in both kernels.  The three cases, and thirty more of every shape, as
    1 / 3 / 15 rows (four wavefronts per utterance), twice each, against the 16-row search (one wavefront per utterance):
    hypotheses, lengths and scores bit for bit."""
    import sys
    from conftest import ROOT
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import soak_beam
    vocabs = soak_beam.make_vocabs(str(tmp_path))
    decs, stats = {}, {"searches": 0, "overflow_rows": 0}
    bad = [m for m in (soak_beam.run_case(seed, vocabs, decs, 2, gpu, stats)
                       for seed in [500628, 500895, 501437] + list(range(730_000, 730_030))) if m]
    assert not bad, bad
    assert stats["overflow_rows"] == 0


@pytest.mark.gpu
def test_beam_search_on_a_long_recording_takes_the_long_transcript_path(gpu):
    """beam_wave.hip assembles a transcript in LDS when the utterance has <= 3 072 frames and goes through HBM (characters
    written from the back of the id row, then moved to its front) beyond that: 3 500 frames (70 s of audio) against the
    oracle, next to the same posteriors cut to 3 000 frames (the LDS path), narrow beam so that the Python oracle finishes."""
    from viet_asr_amd.beam import BeamSearchDecoder
    lp = ctc_like_posteriors(3500, 29, 4242, p_blank=0.7)
    dec = BeamSearchDecoder(LABELS, lm_path=None)
    for T in (3500, 3000):
        x = torch.from_numpy(lp[None, :T].copy()).to(gpu)
        ids, n, score = dec.decode_ids(x, 8)
        text = "".join(LABELS[c] for c in ids[0, : int(n[0])].tolist())
        ref = BO.decode_beams(np.exp(lp[:T].astype(np.float64)), LABELS, 8)
        close = len(ref) > 1 and abs(ref[0][2] - ref[1][2]) < 1e-3
        assert text == ref[0][0] or (close and text == ref[1][0]), (T, text[:60], ref[0][0][:60])
        assert len(text) > 200                      # a transcript long enough to matter
        if text == ref[0][0]:
            assert abs(float(score[0]) - ref[0][2]) < 2e-3 * max(1.0, abs(ref[0][2]) / 50)
