"""Host side of the device prefix beam search (csrc/beam_wave.hip, csrc/beam_group.hip): n-gram tables, launch, id -> text.

Counterpart of what ``BeamSearchDecoderWithLM.__init__`` builds with ``pyctcdecode.build_ctcdecoder(vocab,
kenlm_model_path, alpha, beta)`` (nemo/collections/asr/beam_search_decoder.py:82-87).  The language model is
read from ARPA text; KenLM's binary formats (``*.binary``; the reference's own files are missing anyway,
.MISSING_LARGE_BLOBS:4-7) are a third-party on-disk layout and are refused with a clear error.

pyctcdecode treats the two kinds of file differently, by the path's SUFFIX (``build_ctcdecoder``): for ``*.arpa`` it reads
the file's unigrams, keeps a unigram set and a character trie of it -- a partial word that is a prefix of a known word
carries no out-of-vocabulary penalty, a committed word outside the set gets the unk offset; for anything else there is no
set and every partial word is penalised.  ``unigrams="auto"`` (the default here) does exactly that: ".arpa" => the set
``load_unigram_set_from_arpa`` would read, other suffixes => None.  ``unigrams=None`` asks for the no-unigram behaviour on
any file -- what the reference got from the ``.binary`` its ARPA text was compiled to.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib

_MASK = (1 << 64) - 1
_FNV_PRIME = 1099511628211


def _hstep(h, v):
    return ((h ^ ((v + 1) & _MASK)) * _FNV_PRIME) & _MASK


def read_arpa(path):
    """-> (order, {tuple(words): (log10 p, log10 backoff)})"""
    with open(path, "rb") as f:
        head = f.read(64)
    if not head.lstrip().startswith(b"\\data\\"):
        raise NotImplementedError(f"{path}: only ARPA text n-gram models are supported (KenLM binary formats are not)")
    ngrams, order, cur = {}, 0, 0
    with open(path, encoding="utf-8") as f:
        for line in f:
            line = line.strip()
            if not line or line == "\\data\\" or line.startswith("ngram "):
                continue
            if line == "\\end\\":
                break
            if line.startswith("\\") and line.endswith("-grams:"):
                cur = int(line[1:line.index("-")])
                order = max(order, cur)
                continue
            parts = line.split()
            words = tuple(parts[1:1 + cur])
            ngrams[words] = (float(parts[0]), float(parts[1 + cur]) if len(parts) > 1 + cur else 0.0)
    return order, ngrams


def load_unigram_set_from_arpa(path):
    """The unigram list pyctcdecode's ``build_ctcdecoder`` reads from an ``.arpa`` path (its
    ``language_model.load_unigram_set_from_arpa``): the words of the ``\\1-grams:`` section whose line has exactly three
    TAB-separated fields -- probability, word, back-off.  A 1-gram printed without a back-off weight (KenLM prints ``</s>``
    so, and every word of a unigram-only model) is NOT in the list."""
    unigrams = set()
    with open(path, encoding="utf-8") as f:
        inside = False
        for line in f:
            line = line.strip()
            if line == "\\1-grams:":
                inside = True
            elif line == "\\2-grams:":
                break
            if inside and line:
                parts = line.split("\t")
                if len(parts) == 3:
                    unigrams.add(parts[1])
    if not unigrams:
        raise ValueError(f"{path}: no unigrams found in the ARPA file (pyctcdecode raises here as well)")
    return unigrams


def _trie_buckets(keys):
    """The character trie's node keys as the bucketed set vasr_lm_create() takes (include/vasr.h): 2^lg buckets of two u64
    cells, a key sits in its home bucket ``_home(key, buckets)`` or the next one with a free cell; <= 25 % of the cells used
    (a full bucket -- the only thing that sends a lookup past its first 16-byte load -- is then ~9 % of them)."""
    keys = np.unique(np.asarray(keys, dtype=np.uint64) | np.uint64(1))
    nb = 16
    while 2 * nb < 4 * len(keys):
        nb *= 2
    cells = np.zeros((nb, 2), dtype=np.uint64)
    fill = np.zeros(nb, dtype=np.int64)
    pos = _home(keys, nb) if len(keys) else np.zeros(0, dtype=np.int64)
    todo = np.arange(len(keys))
    while len(todo):                                          # rounds: per bucket the first (free cells) claimants win
        p = pos[todo]
        order = np.argsort(p, kind="stable")
        ps, ts = p[order], todo[order]
        first = np.r_[True, ps[1:] != ps[:-1]]
        start = np.maximum.accumulate(np.where(first, np.arange(len(ps)), 0))
        rank = np.arange(len(ps)) - start                     # position among this round's claimants of the same bucket
        ok = rank < (2 - fill[ps])
        cells[ps[ok], fill[ps[ok]] + rank[ok]] = keys[ts[ok]]
        np.add.at(fill, ps[ok], 1)
        todo = ts[~ok]
        pos[todo] = (pos[todo] + 1) & (nb - 1)
    return cells


def _hstep_np(h, v):
    """_hstep on uint64 arrays (wrap-around arithmetic)."""
    with np.errstate(over="ignore"):
        return (h ^ (v.astype(np.uint64) + np.uint64(1))) * np.uint64(_FNV_PRIME)


def _home(keys, cap):
    """Home slot of a key in a table of 2^lg slots: Fibonacci hashing of the xor-folded key.  (The keys are FNV-style
    products of small word ids; a plain bit field of them clusters badly.)"""
    lg = int(cap).bit_length() - 1
    f = (keys ^ (keys >> np.uint64(32))) & np.uint64(0xFFFFFFFF)
    return (((f * np.uint64(0x9E3779B1)) & np.uint64(0xFFFFFFFF)) >> np.uint64(32 - lg)).astype(np.int64)


def _table(keys, cap):
    """Open-addressing table the kernels probe linearly from ``_home(key, cap)`` (csrc/beam_common.h lm_home): returns
    (slots [cap] u64, position of every key).  Built in vectorised rounds -- a pure-Python insert loop takes minutes on the
    multi-million-entry models the reference ships (.MISSING_LARGE_BLOBS:4-7): in round r every unplaced key looks at
    slot (home + r) & (cap - 1); one key per free slot wins, the others move on.  A key only ever steps over slots that
    are occupied for good, so every probe chain from a key's home to its slot is gap-free, which is all lookup needs."""
    keys = np.asarray(keys, dtype=np.uint64) | np.uint64(1)
    if len(np.unique(keys)) != len(keys):
        raise ValueError("64-bit hash collision while building the n-gram tables")
    slots = np.zeros(cap, dtype=np.uint64)
    where = np.full(len(keys), -1, dtype=np.int64)
    pos = _home(keys, cap)
    todo = np.arange(len(keys))
    while len(todo):
        p = pos[todo]
        free = slots[p] == 0
        cand, cp = todo[free], p[free]
        _, first = np.unique(cp, return_index=True)          # one winner per free slot
        win = cand[first]
        slots[pos[win]] = keys[win]
        where[win] = pos[win]
        todo = todo[where[todo] < 0]
        pos[todo] = (pos[todo] + 1) & (cap - 1)
    return slots, where


def _cap(n):
    """Power-of-two capacity, load <= 50 %."""
    c = 16
    while c < 2 * n + 1:
        c *= 2
    return c


_ENTRY = np.dtype([("key", "<u8"), ("a", "<u4"), ("b", "<u4")])      # 16-byte table entry (include/vasr.h)


class DeviceLM:
    """Uploads an ARPA model as the two hash tables vasr_lm_create() expects."""

    def __init__(self, path, labels, alpha, beta, unk_offset=-10.0, unigrams="auto"):
        """unigrams: "auto" = what build_ctcdecoder does for this path (the ARPA's own list for a "*.arpa" suffix, else none),
        None = no unigram list, or an iterable of words (build_ctcdecoder's ``unigrams=`` argument)."""
        order, ngrams = read_arpa(path)
        if isinstance(unigrams, str):
            if unigrams != "auto":
                raise ValueError("unigrams: 'auto', None or an iterable of words")
            unigrams = load_unigram_set_from_arpa(path) if str(path).endswith(".arpa") else None
        if order > 5:
            raise NotImplementedError("n-gram order > 5")
        L = _lib.lib()
        h0 = int(L.vasr_beam_hash_init())
        assert _hstep(h0, 3) == int(L.vasr_beam_hash_step(h0, 3)), "host/device hash mismatch"
        lab = {c: i for i, c in enumerate(labels)}
        words = sorted({w for ng in ngrams for w in ng})
        for special in ("<s>", "</s>", "<unk>"):
            if special not in words:
                words.append(special)
        wid = {w: i for i, w in enumerate(words)}
        # pyctcdecode LanguageModel.__init__: the unigrams the model knows ("t in kenlm_model": in its vocabulary, not <unk>)
        model_words = {ng[0] for ng in ngrams if len(ng) == 1} - {"<unk>"}
        self.unigram_set = None if unigrams is None else {t for t in set(unigrams) if t in model_words}
        if self.unigram_set is not None and not self.unigram_set:
            self.unigram_set = None          # an empty set behaves exactly like none (score: len(set) > 0; trie: no node)
        # word string -> hash over its label ids; words with characters outside the labels can never be emitted
        vkeys, vvals, vflags = [], [], []
        for w in words:
            if w in ("<s>", "</s>", "<unk>") or any(ch not in lab for ch in w):
                continue
            h = h0
            for ch in w:
                h = _hstep(h, lab[ch])
            vkeys.append(h)
            vvals.append(wid[w])
            vflags.append(1 if self.unigram_set is not None and w in self.unigram_set else 0)
        vcap = _cap(len(vkeys))
        vslots, vwhere = _table(vkeys, vcap)
        vocab = np.zeros(vcap, dtype=_ENTRY)
        vocab["key"] = vslots
        vocab["a"][vwhere] = np.asarray(vvals, dtype=np.int32).view(np.uint32)
        vocab["b"][vwhere] = np.asarray(vflags, dtype=np.uint32)
        # the character trie's nodes: every non-empty prefix of every set member, as far as the labels spell it (has_node of a
        # partial word the search can form)
        trie = None
        if self.unigram_set is not None:
            tkeys = set()
            for w in self.unigram_set:
                h = h0
                for ch in w:
                    if ch not in lab:
                        break
                    h = _hstep(h, lab[ch])
                    tkeys.add(h)
            trie = _trie_buckets(sorted(tkeys))
            self.n_trie_nodes = len(tkeys)
        # n-gram keys: the word ids folded from the LAST word backwards (the keys of every suffix of a history then come
        # out of one chain on the device), one vectorised pass per order
        by_n = {}
        for ng, pb in ngrams.items():
            by_n.setdefault(len(ng), ([], []))
            by_n[len(ng)][0].append([wid[w] for w in ng])
            by_n[len(ng)][1].append(pb)
        nkeys, nvals = [], []
        for n, (ids, pbs) in sorted(by_n.items()):
            ids = np.asarray(ids, dtype=np.uint64).reshape(len(ids), n)
            h = np.full(len(ids), h0, dtype=np.uint64)
            for i in reversed(range(n)):
                h = _hstep_np(h, ids[:, i])
            nkeys.append(h)
            nvals.append(np.asarray(pbs, dtype=np.float32).reshape(len(ids), 2))
        nkeys = np.concatenate(nkeys) if nkeys else np.zeros(0, dtype=np.uint64)
        nvals = np.concatenate(nvals) if nvals else np.zeros((0, 2), dtype=np.float32)
        ncap = _cap(len(nkeys))
        nslots, nwhere = _table(nkeys, ncap)
        ngram = np.zeros(ncap, dtype=_ENTRY)
        ngram["key"] = nslots
        ngram["a"][nwhere] = nvals[:, 0].view(np.uint32)
        ngram["b"][nwhere] = nvals[:, 1].view(np.uint32)
        self.order, self.n_words, self.n_ngrams = order, len(words), len(ngrams)
        self.table_load = len(nkeys) / ncap
        self._h = C.c_void_p()
        _lib.check(L.vasr_lm_create(vocab.ctypes.data, vcap, ngram.ctypes.data, ncap,
                                    trie.ctypes.data if trie is not None else None, len(trie) if trie is not None else 0,
                                    order, wid["<s>"], wid["</s>"], wid["<unk>"], float(alpha), float(beta),
                                    float(unk_offset), C.byref(self._h)))

    @property
    def handle(self):
        return self._h

    def __del__(self):
        try:
            if self._h:
                _lib.lib().vasr_lm_destroy(self._h)
        except Exception:
            pass


def lm_file_usable(path):
    """True for an ARPA text model; False (with the reason) for anything else, e.g. KenLM's binary formats."""
    try:
        with open(path, "rb") as f:
            head = f.read(64)
    except OSError as e:
        return False, str(e)
    if not head.lstrip().startswith(b"\\data\\"):
        return False, "only ARPA text n-gram models are supported (KenLM binary formats are a third-party layout)"
    return True, ""


LM_HELP = ("this library reads n-gram models as ARPA text only (or put `<same stem>.arpa` next to the `.binary`: it is then read "
           "instead, with the binary's no-unigram behaviour).  The reference's default `models/language_model/3-gram-lm.binary` "
           "(infer.py:184, app.py:20) is a KenLM binary, a third-party layout that cannot be turned back into ARPA: keep (or "
           "rebuild) the ARPA file it was compiled from -- `lmplz -o 3 < corpus.txt > 3-gram-lm.arpa` -- and pass that path.  "
           "Pass allow_missing_lm=True to search without a language model instead.")


class BeamSearchDecoder:
    def __init__(self, labels, lm_path=None, alpha=0.5, beta=1.5, token_min_logp=-5.0, beam_prune_logp=-10.0,
                 allow_missing_lm=False, unigrams="auto"):
        self.labels = list(labels)
        if len(self.labels) + 1 > 128:
            raise NotImplementedError("beam search supports at most 127 labels + blank")
        self.space_id = self.labels.index(" ") if " " in self.labels else -1
        self.token_min_logp, self.beam_prune_logp = token_min_logp, beam_prune_logp
        # An LM that cannot be used is decided HERE, not at the first decode, and it is an ERROR (round 5): a drop-in user
        # with the reference's shipped paths must not silently lose the language model.  allow_missing_lm=True restores the
        # reference's own fall-back (infer.py:117-128: kenlm not importable -> lm_path = None, search without an LM).
        if lm_path:
            ok, why = lm_file_usable(lm_path)
            if not ok:
                # The reference's shipped paths name KenLM binaries (`models/language_model/3-gram-lm.binary`, infer.py:184,
                # app.py:20).  When the ARPA text such a binary was compiled from sits NEXT to it under the same stem, read that
                # -- with the behaviour the BINARY has in pyctcdecode (no unigram list), so that an unmodified call site gets the
                # rankings it would get from the reference.
                import os
                sibling = lm_path.rsplit(".", 1)[0] + ".arpa" if lm_path.endswith((".binary", ".bin")) else None
                if sibling and os.path.exists(sibling) and lm_file_usable(sibling)[0]:
                    import warnings
                    warnings.warn(f"language model {lm_path!r} not usable ({why}); reading its ARPA source {sibling!r} instead, with the "
                                  "behaviour a KenLM binary has in pyctcdecode (no unigram list)")
                    lm_path, ok = sibling, True
                    if isinstance(unigrams, str) and unigrams == "auto":
                        unigrams = None
            if not ok:
                if not allow_missing_lm:
                    raise ValueError(f"language model {lm_path!r} not usable ({why}); {LM_HELP}")
                import warnings
                warnings.warn(f"language model {lm_path!r} not usable ({why}); beam search runs without a language model")
                lm_path = None
        self.lm_path, self.alpha, self.beta = lm_path, alpha, beta
        # "auto": pyctcdecode's rule for the path's suffix (module docstring); None: no unigram list; or a list of words
        self.unigrams = unigrams
        self._lm = None
        self._ws = {}
        self._ws_lock = __import__("threading").Lock()

    def _get_lm(self):
        if self.lm_path and self._lm is None:
            self._lm = DeviceLM(self.lm_path, self.labels, self.alpha, self.beta, unigrams=self.unigrams)
        return self._lm

    def decode_ids(self, log_probs, beam_width, frames=None):
        """log_probs [B,T,V+1] cuda f32 -> (ids [B,T] i32, id_len [B] i32, score [B] f32).

        frames: optional [B] frame counts (sequence or tensor); row b is then searched over its first frames[b]
        frames only -- for batches of different lengths (the reference searches every frame of its batch-1 tensor)."""
        if log_probs.device.type != "cuda":
            raise _lib.VasrError("viet-asr_amd kernels need HIP-resident tensors; there is no CPU fallback for this path")
        x = log_probs.to(torch.float32).contiguous()
        B, T, V1 = x.shape
        if V1 != len(self.labels) + 1:
            raise ValueError(f"log_probs has {V1} classes, expected {len(self.labels)} labels + blank")
        L = _lib.lib()
        need = int(L.vasr_beam_workspace_bytes(B, T))
        stream = torch.cuda.current_stream().cuda_stream
        with self._ws_lock:                         # one workspace per (device, stream): two threads on two streams do not share one
            ws = self._ws.get((x.device, stream))
            if ws is None or ws.numel() < need:
                ws = self._ws[(x.device, stream)] = torch.empty(need, dtype=torch.uint8, device=x.device)
        ids = torch.empty((B, T), dtype=torch.int32, device=x.device)
        n = torch.empty((B,), dtype=torch.int32, device=x.device)
        score = torch.empty((B,), dtype=torch.float32, device=x.device)
        lm = self._get_lm()
        rows = None
        if frames is not None:
            rows = torch.as_tensor(frames).to(device=x.device, dtype=torch.int32).contiguous()
            if rows.shape != (B,):
                raise ValueError(f"frames must have one entry per row ({B}), got shape {tuple(rows.shape)}")
        _lib.check(L.vasr_beam_search_rows_f32(x.data_ptr(), rows.data_ptr() if rows is not None else None, B, T, V1,
                                               self.space_id, int(beam_width), float(self.token_min_logp),
                                               float(self.beam_prune_logp), lm.handle if lm is not None else None,
                                               ids.data_ptr(), n.data_ptr(), score.data_ptr(), ws.data_ptr(),
                                               ws.numel(), stream))
        return ids, n, score

    def decode_batch(self, log_probs, beam_width, frames=None):
        ids, n, _ = self.decode_ids(log_probs, beam_width, frames)
        ids, n = ids.cpu().numpy(), n.cpu().numpy()
        if (n < 0).any():      # vasr.h: id_len = -1 reports a merge-cell overflow of the four-wavefront kernel (provably impossible)
            raise _lib.VasrError("beam search reported an internal overflow (id_len = -1) for rows %s" % np.nonzero(n < 0)[0].tolist())
        return ["".join(self.labels[c] for c in ids[b, : n[b]]) for b in range(ids.shape[0])]
