"""CPU oracle for BeamSearchDecoderWithLM -- TEST INFRASTRUCTURE ONLY (see quartznet_oracle.py header).

*** PARITY UNPINNED ***  The reference's beam search is arithmetic in third-party packages that are neither
vendored in /root/reference nor installed in this image: ``pyctcdecode`` (requirements.txt:16, unpinned) on top of
``kenlm`` (README.md:43-45, GitHub master); the LM binaries are missing as well (.MISSING_LARGE_BLOBS:4-7).  The
reference holds no test or golden vector for this path.  What follows restates pyctcdecode's *published*
algorithm (v0.5.0 ``BeamSearchDecoderCTC._decode_logits`` / ``_get_lm_beams`` / ``LanguageModel.score``) anchored
on the reference's call site:

    nemo/collections/asr/beam_search_decoder.py:82-87   build_ctcdecoder(vocab, kenlm_model_path, alpha, beta)
    nemo/collections/asr/beam_search_decoder.py:95-102  probs = exp(log_probs[0]); decoder.decode(probs, beam_width)

so: no hotwords, default ``beam_prune_logp=-10``, ``token_min_logp=-5``, ``unk_score_offset=-10``,
``prune_history=False``; the blank is the LAST class (pyctcdecode appends "" to the labels).  Scores are natural-log; LM
scores are ``alpha * log10_score * ln(10) + beta`` per scored word.

TWO behaviours hide behind that one call, chosen by the SUFFIX of ``kenlm_model_path`` ("either .arpa or .bin file",
beam_search_decoder.py:84) -- ``build_ctcdecoder`` as published:

    unigrams is None and path.endswith(".arpa")  ->  unigrams = load_unigram_set_from_arpa(path)
    otherwise (the reference's own ``3-gram-lm.binary``, infer.py:184)  ->  unigrams stay None (a warning)

* ``unigrams=None``  ("binary" semantics): no character trie, EVERY partial word is "OOV" (``is_oov = 1.0``); a committed word
  gets the unk offset iff it is not in the n-gram model's vocabulary.
* unigrams given  ("arpa" semantics): ``unigram_set = {t in unigrams if t in kenlm_model}``, ``char_trie =
  CharTrie.fromkeys(unigram_set)``; a partial word is OOV iff ``char_trie.has_node(partial) == 0`` -- i.e. iff it is NOT a prefix
  of (or equal to) a word of the set; a committed word gets the unk offset iff ``word not in unigram_set or word not in
  kenlm_model``.  ``load_unigram_set_from_arpa`` keeps the 1-gram lines that split into exactly THREE tab-separated fields
  (probability, word, back-off): a unigram printed without a back-off weight (KenLM prints ``</s>`` and, in a unigram-only
  model, every word that way) is in the model but NOT in the set.
``LanguageModel(unigrams=...)`` / ``load_unigram_set_from_arpa`` / ``unigrams_for_path`` below restate exactly that; the
prefix set ``_prefixes`` stands in for ``pygtrie.CharTrie.has_node``.

The n-gram model is a plain back-off model read from ARPA text (``NgramLM``): KenLM's ``BaseScore(state, word)``
on a full (order-1)-word history, which is what KenLM computes (its state minimisation does not change scores).
"""
import math

import numpy as np

MIN_TOKEN_CLIP_P = 1e-15
LOG_BASE_CHANGE_FACTOR = 1.0 / math.log10(math.e)
AVG_TOKEN_LEN = 6
DEFAULT_BEAM_PRUNE_LOGP = -10.0
DEFAULT_TOKEN_MIN_LOGP = -5.0
DEFAULT_UNK_LOGP_OFFSET = -10.0


class NgramLM:
    """Back-off n-gram LM from ARPA text; log10 probabilities; ``<s>``, ``</s>``, ``<unk>`` as in KenLM."""

    def __init__(self, order, ngrams):
        self.order = order
        self.ngrams = ngrams                  # {tuple(words): (log10 prob, log10 backoff)}
        self.vocab = {w[0] for w in ngrams if len(w) == 1}

    @classmethod
    def from_arpa(cls, path):
        ngrams, order, cur = {}, 0, 0
        with open(path, encoding="utf-8") as f:
            for line in f:
                line = line.strip()
                if not line or line.startswith("ngram ") or line == "\\data\\":
                    continue
                if line.startswith("\\") and line.endswith("-grams:"):
                    cur = int(line[1:line.index("-")])
                    order = max(order, cur)
                    continue
                if line == "\\end\\":
                    break
                parts = line.split("\t") if "\t" in line else line.split()
                if "\t" in line:
                    prob, words = float(parts[0]), tuple(parts[1].split())
                    bo = float(parts[2]) if len(parts) > 2 else 0.0
                else:
                    prob, words = float(parts[0]), tuple(parts[1:1 + cur])
                    bo = float(parts[1 + cur]) if len(parts) > 1 + cur else 0.0
                ngrams[words] = (prob, bo)
        return cls(order, ngrams)

    def __contains__(self, word):       # kenlm.Model.__contains__: vocabulary index != 0, and index 0 is <unk>
        return word in self.vocab and word != "<unk>"

    def begin_state(self):
        return ("<s>",)

    def base_score(self, state, word):
        """log10 p(word | state) with back-off; returns (score, new_state)."""
        w = word if word in self.vocab else "<unk>"
        ctx = tuple(state)[-(self.order - 1):] if self.order > 1 else ()
        score = 0.0
        while True:
            hit = self.ngrams.get(ctx + (w,))
            if hit is not None:
                score += hit[0]
                break
            if not ctx:
                score += self.ngrams.get(("<unk>",), (-100.0, 0.0))[0]
                break
            score += self.ngrams.get(ctx, (0.0, 0.0))[1]
            ctx = ctx[1:]
        new_state = (tuple(state) + (w,))[-(self.order - 1):] if self.order > 1 else ()
        return score, new_state


def load_unigram_set_from_arpa(arpa_path):
    """pyctcdecode.language_model.load_unigram_set_from_arpa as published: the words of the ``\\1-grams:`` section whose line
    splits into exactly three TAB-separated fields; raises when none is found."""
    unigrams = set()
    with open(arpa_path, encoding="utf-8") as f:
        start_1_gram = False
        for line in f:
            line = line.strip()
            if line == "\\1-grams:":
                start_1_gram = True
            elif line == "\\2-grams:":
                break
            if start_1_gram and len(line) > 0:
                parts = line.split("\t")
                if len(parts) == 3:
                    unigrams.add(parts[1])
    if len(unigrams) == 0:
        raise ValueError("No unigrams found in arpa file. Something is wrong with the file.")
    return unigrams


def unigrams_for_path(kenlm_model_path):
    """What build_ctcdecoder(labels, kenlm_model_path, alpha, beta) -- the reference's call, no ``unigrams`` argument --
    ends up with: the ARPA's own unigram list for a path that ENDS in ".arpa", None for anything else."""
    if kenlm_model_path is not None and kenlm_model_path.endswith(".arpa"):
        return load_unigram_set_from_arpa(kenlm_model_path)
    return None


class LanguageModel:
    """pyctcdecode.language_model.LanguageModel.  ``unigrams=None`` is what build_ctcdecoder leaves for a ``.binary`` path,
    a collection of words what it loads for an ``.arpa`` path (file header)."""

    def __init__(self, lm, alpha=0.5, beta=1.5, unk_score_offset=DEFAULT_UNK_LOGP_OFFSET, score_boundary=True,
                 unigrams=None):
        self.lm, self.alpha, self.beta, self.unk_score_offset, self.score_boundary = lm, alpha, beta, unk_score_offset, score_boundary
        if unigrams is None:
            self._unigram_set, self._prefixes = set(), None
        else:
            self._unigram_set = {t for t in set(unigrams) if t in lm}
            # CharTrie.fromkeys(unigram_set).has_node(s)  <=>  s is a prefix of (or equal to) a key
            self._prefixes = {w[:i] for w in self._unigram_set for i in range(1, len(w) + 1)}

    def get_start_state(self):
        return self.lm.begin_state() if self.score_boundary else ()

    def score_partial_token(self, partial_token):
        if self._prefixes is None:
            is_oov = 1.0                                  # no char trie
        else:
            is_oov = int(partial_token not in self._prefixes)
        unk_score = self.unk_score_offset * is_oov
        if len(partial_token) > AVG_TOKEN_LEN:
            unk_score = unk_score * len(partial_token) / AVG_TOKEN_LEN
        return unk_score

    def score(self, prev_state, word, is_last_word=False):
        lm_score, end_state = self.lm.base_score(prev_state, word)
        if (len(self._unigram_set) > 0 and word not in self._unigram_set) or word not in self.lm:
            lm_score += self.unk_score_offset
        if is_last_word and self.score_boundary:
            lm_score += self.lm.base_score(end_state, "</s>")[0]
        return self.alpha * lm_score * LOG_BASE_CHANGE_FACTOR + self.beta, end_state


def _sum_log_scores(s1, s2):
    if s1 >= s2:
        return s1 + math.log(1 + math.exp(s2 - s1))
    return s2 + math.log(1 + math.exp(s1 - s2))


def _merge_tokens(a, b):
    if not b:
        return a
    return b if not a else a + " " + b


def _merge_beams(beams):
    d = {}
    for text, next_word, word_part, last_char, logit_score in beams:
        key = (_merge_tokens(text, next_word), word_part, last_char)
        if key not in d:
            d[key] = (text, next_word, word_part, last_char, logit_score)
        else:
            d[key] = (text, next_word, word_part, last_char, _sum_log_scores(d[key][-1], logit_score))
    return list(d.values())


def _lm_beams(beams, lm, cached_lm, cached_partial, is_eos=False):
    out = []
    for text, next_word, word_part, last_char, logit_score in beams:
        new_text = _merge_tokens(text, next_word)
        if lm is None:
            out.append((new_text, "", word_part, last_char, logit_score, logit_score))
            continue
        if new_text not in cached_lm:
            _, prev_raw, start_state = cached_lm[text]
            score, end_state = lm.score(start_state, next_word, is_last_word=is_eos)
            cached_lm[new_text] = (prev_raw + score, prev_raw + score, end_state)
        lm_score = cached_lm[new_text][0]
        if word_part:
            if word_part not in cached_partial:
                cached_partial[word_part] = lm.score_partial_token(word_part)
            lm_score += cached_partial[word_part]
        out.append((new_text, "", word_part, last_char, logit_score, logit_score + lm_score))
    return out


def decode_beams(probs, labels, beam_width, lm=None, beam_prune_logp=DEFAULT_BEAM_PRUNE_LOGP,
                 token_min_logp=DEFAULT_TOKEN_MIN_LOGP):
    """probs [T, V+1] (rows sum to 1, blank last) -> list of (text, logit_score, combined_score), best first.

    pyctcdecode's loop as published, nothing bent towards the device kernel: every character that clears
    ``token_min_logp`` is expanded on every beam, and the final </s> pass goes through the LM score cache the frames
    filled (a text whose commit was scored on an earlier frame keeps that cached score, without </s>)."""
    probs = np.asarray(probs, dtype=np.float64)
    logits = np.log(np.clip(probs, MIN_TOKEN_CLIP_P, 1))          # rows look like probabilities
    idx2vocab = list(labels) + [""]
    cached_lm = {"": (0.0, 0.0, lm.get_start_state())} if lm is not None else {}
    cached_partial = {}
    beams = [("", "", "", None, 0.0)]
    for col in logits:
        idx_list = set(np.where(col >= token_min_logp)[0]) | {int(col.argmax())}
        new_beams = []
        for idx in sorted(idx_list):
            p_char, char = col[idx], idx2vocab[idx]
            for text, next_word, word_part, last_char, logit_score in beams:
                if char == "" or last_char == char:
                    new_beams.append((text, next_word, word_part, char, logit_score + p_char))
                elif char == " ":
                    new_beams.append((text, word_part, "", char, logit_score + p_char))
                else:
                    new_beams.append((text, next_word, word_part + char, char, logit_score + p_char))
        scored = _lm_beams(_merge_beams(new_beams), lm, cached_lm, cached_partial)
        max_score = max(b[-1] for b in scored)
        scored = [b for b in scored if b[-1] >= max_score + beam_prune_logp]
        scored.sort(key=lambda b: -b[-1])
        beams = [b[:-1] for b in scored[:beam_width]]
    final = [(text, word_part, "", None, logit_score) for text, _, word_part, _, logit_score in beams]
    scored = _lm_beams(_merge_beams(final), lm, cached_lm, cached_partial, is_eos=True)
    max_score = max(b[-1] for b in scored)
    scored = [b for b in scored if b[-1] >= max_score + beam_prune_logp]
    scored.sort(key=lambda b: -b[-1])
    return [(" ".join(b[0].split()), b[-2], b[-1]) for b in scored[:beam_width]]


def decode(log_probs_row, labels, beam_width, lm=None, **kw):
    """BeamSearchDecoderWithLM.forward for one utterance: log_probs [T, V+1] -> best text."""
    return decode_beams(np.exp(np.asarray(log_probs_row, dtype=np.float64)), labels, beam_width, lm=lm, **kw)[0][0]


def write_arpa(path, order, ngrams, no_backoff=()):
    """ngrams: {tuple(words): (log10 p, log10 backoff)} -> ARPA text (test helper).  Unigrams named in ``no_backoff`` are
    printed without the back-off field, as KenLM prints ``</s>``: in the model, not in pyctcdecode's unigram set."""
    with open(path, "w", encoding="utf-8") as f:
        f.write("\\data\\\n")
        for n in range(1, order + 1):
            f.write(f"ngram {n}={sum(1 for w in ngrams if len(w) == n)}\n")
        for n in range(1, order + 1):
            f.write(f"\n\\{n}-grams:\n")
            for w, (p, bo) in sorted(ngrams.items()):
                if len(w) == n:
                    f.write(f"{p:.6f}\t{' '.join(w)}" + (f"\t{bo:.6f}" if n < order and not (n == 1 and w[0] in no_backoff) else "") + "\n")
        f.write("\n\\end\\\n")
