// Device-side pieces shared by the beam-search kernels (beam_wave.hip: one wavefront per utterance, batches;
// beam_group.hip: an utterance on four wavefronts of a compute unit, the serving latency): hashing,
// order-preserving score bits, the hashed back-off n-gram model (KenLM BaseScore semantics, pyctcdecode's LanguageModel.score
// on top), log(r >= 1) in fp64 without the library call, and wavefront-wide scans / reductions on the DPP data path.
#pragma once
#include <hip/hip_runtime.h>

#include "vasr_internal.h"

namespace vasr {
namespace beam_detail {

constexpr int kMaxCtx = 4;      // LM order <= 5
constexpr int kMaxClasses = 128;
constexpr int kMaxBeams = 128;
constexpr unsigned long long kFnvOffset = 1469598103934665603ull, kFnvPrime = 1099511628211ull;
constexpr double kFix = 17592186044416.0;  // 2^44

__host__ __device__ inline unsigned long long hmix(unsigned long long h, unsigned long long v) {
  return (h ^ (v + 1)) * kFnvPrime;
}
__device__ inline long long ord64(double d) {  // order-preserving map double -> signed 64
  long long b = __double_as_longlong(d);
  return b >= 0 ? b : (long long)(0x8000000000000000ull ^ (unsigned long long)~b) ;
}
__device__ inline double unord64(long long o) {
  long long b = o >= 0 ? o : (long long)~(0x8000000000000000ull ^ (unsigned long long)o);
  return __longlong_as_double(b);
}

// The n-gram model on the device (vasr_lm_create, include/vasr.h): two open-addressing tables of 16-BYTE entries in
// HBM -- one load returns key and value -- with power-of-two capacities 2^lg, linear probing from the home slot
// lm_home(key, lg) = ((u32)(key ^ key >> 32) * 0x9E3779B1) >> (32 - lg)  (the keys are FNV-style products of small word ids:
// their entropy sits in bits 0-24 and 40+, any plain bit field of them clusters -- a first version that used bits 17.. sent
// 20 000 unigrams to 256 home slots and probe chains ran to thousands of entries):
//   vocabulary  {u64 key = hash of the word's label ids | 1, i32 word id, u32 flags: bit 0 = member of pyctcdecode's unigram set}
//   n-grams     {u64 key | 1, f32 log10 p, f32 log10 back-off}; the key of (w_1 .. w_n) is folded from the LAST word
//               backwards, key = hmix(... hmix(hmix(offset, w_n), w_{n-1}) ..., w_1): the keys of all suffixes of a history
//               come out of one chain, and a back-off walk needs every one of them.
// and, when the decoder was built with a unigram list (pyctcdecode's behaviour for an ".arpa" path: LanguageModel's
// unigram_set + CharTrie), the NODES of the character trie as a set of 64-bit keys:
//   trie        2^tlg buckets of TWO u64 keys (16 bytes, one load): key = hash of a word PREFIX's label ids | 1 -- the rolling
//               hash every beam already carries for its pending word -- for every non-empty prefix of every word of the
//               unigram set; home bucket lm_home(key, tlg), linear probing over buckets, a bucket with a free cell ends the walk.
struct LmView {
  const uint4* vocab; int vcap, vlg;      // capacity 2^vlg
  const uint4* ngram; int ncap, nlg;
  const ulonglong2* trie; int tlg;        // trie == nullptr: no unigram list (every partial word is "OOV")
  int order, bos, eos, unk;
  float alpha, beta, unk_offset;
};

inline LmView make_lm_view(const BeamLm* lm) {
  LmView v{};
  if (!lm) return v;
  v.vocab = static_cast<const uint4*>(lm->vocab); v.vcap = lm->vcap; v.ngram = static_cast<const uint4*>(lm->ngram);
  v.ncap = lm->ncap; v.order = lm->order; v.bos = lm->bos;
  v.vlg = 31 - __builtin_clz((unsigned)lm->vcap); v.nlg = 31 - __builtin_clz((unsigned)lm->ncap);
  v.trie = static_cast<const ulonglong2*>(lm->trie);
  v.tlg = lm->trie ? 31 - __builtin_clz((unsigned)lm->tbuckets) : 0;
  v.eos = lm->eos; v.unk = lm->unk; v.alpha = lm->alpha; v.beta = lm->beta; v.unk_offset = lm->unk_offset;
  return v;
}

__device__ inline unsigned long long entry_key(const uint4& e) { return ((unsigned long long)e.y << 32) | e.x; }
__host__ __device__ inline int lm_home(unsigned long long k, int lg) {
  return (int)((((unsigned)k ^ (unsigned)(k >> 32)) * 0x9E3779B1u) >> (32 - lg));
}

// word id of a committed word (-1: not in the n-gram model's vocabulary); *in_set: member of the unigram set (entry flags bit 0)
__device__ __forceinline__ int lm_word_id(const LmView& lm, unsigned long long whash, bool* in_set) {
  const unsigned long long k = whash | 1ull;
  *in_set = false;
  for (int i = lm_home(k, lm.vlg), n = 0; n < lm.vcap; ++n, i = (i + 1) & (lm.vcap - 1)) {
    const uint4 e = lm.vocab[i];
    const unsigned long long ek = entry_key(e);
    if (ek == k) { *in_set = (e.w & 1u) != 0u; return (int)e.z; }
    if (ek == 0) break;
  }
  return -1;  // out of vocabulary
}

// pygtrie CharTrie.has_node(partial word) on the key set: `first` is the home bucket, requested by the caller long before
// the answer is needed (the expand step issues it, the score step -- a table phase and, in the four-wavefront kernel, a
// barrier later -- consumes it); only a FULL bucket that holds neither the key nor a free cell sends the lane on (the next
// bucket is 16 bytes further: as a rule the same cache line).
__device__ __forceinline__ ulonglong2 trie_first(const LmView& lm, unsigned long long whash) {
  return lm.trie[lm_home(whash | 1ull, lm.tlg)];
}
__device__ __forceinline__ bool trie_has_node(const LmView& lm, unsigned long long whash, ulonglong2 first) {
  const unsigned long long k = whash | 1ull;
  const int nb = 1 << lm.tlg;
  int i = lm_home(k, lm.tlg);
  ulonglong2 e = first;
  for (int n = 0; n < nb; ++n) {
    if (e.x == k || e.y == k) return true;
    if (e.x == 0ull || e.y == 0ull) return false;
    i = (i + 1) & (nb - 1);
    e = lm.trie[i];
  }
  return false;
}

// the rest of a probe chain whose first entry `e` was neither the key nor empty (rare at <= 50 % load)
__device__ __forceinline__ bool lm_probe_on(const LmView& lm, unsigned long long k, uint4 e, float2* out) {
  int i = lm_home(k, lm.nlg);
  for (int c = 0; c < lm.ncap; ++c) {
    const unsigned long long ek = entry_key(e);
    if (ek == k) { *out = make_float2(__uint_as_float(e.z), __uint_as_float(e.w)); return true; }
    if (ek == 0) return false;
    i = (i + 1) & (lm.ncap - 1);
    e = lm.ngram[i];
  }
  return false;
}

// KenLM BaseScore on a full history: log10 p(w | ctx) with back-off.  The walk "longest n-gram, else back-off weight of
// its context + the next shorter one" needs the entries of (ctx[s..], w) and of (ctx[s..]) for every start s: their
// keys come from two incremental chains and ALL first probes are requested before any is looked at -- one trip to
// L2 / HBM for the whole walk instead of one per step (ten dependent trips for a trigram model whose words are unseen
// together; the LM was 3 100 of the 14 600 cycles of an average frame).
__device__ __forceinline__ float lm_base_score(const LmView& lm, const int* ctx, int w) {
  int ids[kMaxCtx];               // usable history, most recent LAST
  int n = 0;
  for (int i = 0; i < kMaxCtx; ++i)
    if (ctx[i] >= 0 && kMaxCtx - i <= lm.order - 1) ids[n++] = ctx[i];
  // kf[j]: key of (ids[n-j .. n-1], w), j = 0 .. n (j history words);  kc[j]: key of (ids[n-j .. n-1]), j = 1 .. n
  unsigned long long kf[kMaxCtx + 1], kc[kMaxCtx + 1];
  uint4 ef[kMaxCtx + 1], ec[kMaxCtx + 1];
  unsigned long long hf = hmix(kFnvOffset, (unsigned long long)w), hc = kFnvOffset;
  kf[0] = hf | 1ull;
  kc[0] = 0;
#pragma unroll
  for (int j = 1; j <= kMaxCtx; ++j) {
    if (j <= n) {
      const unsigned long long id = (unsigned long long)ids[n - j];
      hf = hmix(hf, id); hc = hmix(hc, id);
      kf[j] = hf | 1ull; kc[j] = hc | 1ull;
    } else { kf[j] = 0; kc[j] = 0; }
  }
#pragma unroll
  for (int j = 0; j <= kMaxCtx; ++j) {
    ef[j] = make_uint4(0, 0, 0, 0); ec[j] = make_uint4(0, 0, 0, 0);
    if (j <= n) ef[j] = lm.ngram[lm_home(kf[j], lm.nlg)];
    if (j >= 1 && j <= n) ec[j] = lm.ngram[lm_home(kc[j], lm.nlg)];
  }
  float score = 0.f;
  bool done = false;
#pragma unroll
  for (int j = kMaxCtx; j >= 0; --j) {          // longest first
    if (j > n || done) continue;
    float2 v;
    if (lm_probe_on(lm, kf[j], ef[j], &v)) { score += v.x; done = true; continue; }
    if (j == 0) break;
    if (lm_probe_on(lm, kc[j], ec[j], &v)) score += v.y;     // back-off weight of the context
  }
  if (!done) {   // unigram missing: fall back to <unk>
    const unsigned long long ku = hmix(kFnvOffset, (unsigned long long)lm.unk) | 1ull;
    float2 v;
    if (lm_probe_on(lm, ku, lm.ngram[lm_home(ku, lm.nlg)], &v)) score += v.x; else score += -100.f;
  }
  return score;
}

// pyctcdecode LanguageModel.score (alpha * log10 * ln10 + beta, unk offset, optional </s>).  The unk offset: "word not in
// kenlm_model", and -- with a (non-empty) unigram set -- also "word not in unigram_set" (a word of the model whose 1-gram line
// carries no back-off weight is outside the set pyctcdecode reads from the ARPA file: oracle/beam_oracle.py header)
__device__ __forceinline__ float lm_word_score(const LmView& lm, const int* ctx, unsigned long long whash, bool eos, int* wid_out) {
  bool in_set;
  int wid = lm_word_id(lm, whash, &in_set);
  const bool oov = wid < 0;
  if (oov) wid = lm.unk;
  float s = lm_base_score(lm, ctx, wid);
  if (oov || (lm.trie != nullptr && !in_set)) s += lm.unk_offset;
  if (eos) {
    int c2[kMaxCtx];
    for (int i = 0; i < kMaxCtx - 1; ++i) c2[i] = ctx[i + 1];
    c2[kMaxCtx - 1] = wid;
    s += lm_base_score(lm, c2, lm.eos);
  }
  *wid_out = wid;
  return lm.alpha * s * 2.302585092994046f + lm.beta;
}

// pyctcdecode LanguageModel.score_partial_token: unk_offset * is_oov, stretched by len / 6 beyond six characters.  is_oov: 1.0
// without a character trie, else "the partial word is not a node of it" (the beams carry that as a meta bit)
__device__ inline float partial_penalty(float unk_offset, int wlen, bool is_oov) {
  if (wlen <= 0 || !is_oov) return 0.f;
  float u = unk_offset;
  if (wlen > 6) u = u * (float)wlen / 6.0f;
  return u;
}

// log(r) for r in [1, 2^20): exponent split + atanh series (10 odd terms at |s| <= 0.1716: < 1e-16 relative).
// The library log costs ~2400 cycles per wavefront here, every merged prefix needs one per frame.
__device__ inline double log_ge1(double r) {
  long long bits = __double_as_longlong(r);
  int e = (int)((bits >> 52) & 0x7ff) - 1023;
  double m = __longlong_as_double((bits & 0x000fffffffffffffll) | 0x3ff0000000000000ll);   // [1, 2)
  if (m > 1.4142135623730951) { m *= 0.5; e += 1; }                                         // [0.7071, 1.4142]
  // 1/(m+1): hardware estimate + two Newton steps (full IEEE division is ~25 fp64 instructions, 8 cycles each)
  const double d = m + 1.0;
  double r1 = __builtin_amdgcn_rcp(d);
  r1 = fma(fma(-d, r1, 1.0), r1, r1);
  r1 = fma(fma(-d, r1, 1.0), r1, r1);
  const double s = (m - 1.0) * r1, z = s * s;
  double p = 1.0 / 21.0;
  p = fma(p, z, 1.0 / 19.0); p = fma(p, z, 1.0 / 17.0); p = fma(p, z, 1.0 / 15.0); p = fma(p, z, 1.0 / 13.0);
  p = fma(p, z, 1.0 / 11.0); p = fma(p, z, 1.0 / 9.0); p = fma(p, z, 1.0 / 7.0); p = fma(p, z, 1.0 / 5.0);
  p = fma(p, z, 1.0 / 3.0); p = fma(p, z, 1.0);
  return fma((double)e, 0.6931471805599453, 2.0 * s * p);
}

// Wavefront-wide inclusive scan / max on the DPP data path (row shifts + row broadcasts, ~20 VALU instructions).
// The __shfl_up / __shfl_xor forms go through ds_bpermute: six dependent LDS round trips, ~700 cycles per scan, and
// the frame loop runs five or more of them.
template <int CTRL>
__device__ inline int dpp_mov(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false); }

__device__ inline int wave_scan_incl(int v) {
  const int lane = threadIdx.x & 63, rl = lane & 15;
  int x = v, t;
  t = dpp_mov<0x111>(x); if (rl >= 1) x += t;              // row_shr:1
  t = dpp_mov<0x112>(x); if (rl >= 2) x += t;              // row_shr:2
  t = dpp_mov<0x114>(x); if (rl >= 4) x += t;              // row_shr:4
  t = dpp_mov<0x118>(x); if (rl >= 8) x += t;              // row_shr:8
  t = dpp_mov<0x142>(x); if ((lane & 31) >= 16) x += t;    // row_bcast:15
  t = dpp_mov<0x143>(x); if (lane >= 32) x += t;           // row_bcast:31
  return x;
}

__device__ inline unsigned wave_max_u32(unsigned v) {
  const int lane = threadIdx.x & 63, rl = lane & 15;
  unsigned x = v, t;
  t = (unsigned)dpp_mov<0x111>((int)x); if (rl >= 1) x = max(x, t);
  t = (unsigned)dpp_mov<0x112>((int)x); if (rl >= 2) x = max(x, t);
  t = (unsigned)dpp_mov<0x114>((int)x); if (rl >= 4) x = max(x, t);
  t = (unsigned)dpp_mov<0x118>((int)x); if (rl >= 8) x = max(x, t);
  t = (unsigned)dpp_mov<0x142>((int)x); if ((lane & 31) >= 16) x = max(x, t);
  t = (unsigned)dpp_mov<0x143>((int)x); if (lane >= 32) x = max(x, t);
  return (unsigned)__builtin_amdgcn_readlane((int)x, 63);
}

__device__ inline unsigned wave_or_u32(unsigned v) {
  const int lane = threadIdx.x & 63, rl = lane & 15;
  unsigned x = v, t;
  t = (unsigned)dpp_mov<0x111>((int)x); if (rl >= 1) x |= t;
  t = (unsigned)dpp_mov<0x112>((int)x); if (rl >= 2) x |= t;
  t = (unsigned)dpp_mov<0x114>((int)x); if (rl >= 4) x |= t;
  t = (unsigned)dpp_mov<0x118>((int)x); if (rl >= 8) x |= t;
  t = (unsigned)dpp_mov<0x142>((int)x); if ((lane & 31) >= 16) x |= t;
  t = (unsigned)dpp_mov<0x143>((int)x); if (lane >= 32) x |= t;
  return (unsigned)__builtin_amdgcn_readlane((int)x, 63);
}

// max of a signed 64-bit value over the wavefront: high words first, then the low words of the lanes that tie
__device__ inline long long wave_max_i64(long long v) {
  const unsigned long long u = (unsigned long long)v ^ 0x8000000000000000ull;
  const unsigned hi = (unsigned)(u >> 32), lo = (unsigned)u;
  const unsigned hmax = wave_max_u32(hi);
  const unsigned lmax = wave_max_u32(hi == hmax ? lo : 0u);
  return (long long)((((unsigned long long)hmax << 32) | lmax) ^ 0x8000000000000000ull);
}


}  // namespace beam_detail
}  // namespace vasr
