// CTC prefix beam search with optional back-off n-gram LM: ONE UTTERANCE ON FOUR WAVEFRONTS OF ONE COMPUTE UNIT (gfx950) --
// the latency form of beam_wave.hip, for the shape the reference SERVES: batch 1, beam width 50 (app.py:27) or 100
// (infer.py:191) + LM, reference nemo/collections/asr/beam_search_decoder.py:95-102 (pyctcdecode on the host, one utterance
// at a time; third-party, parity unpinned: the algorithm restated is oracle/beam_oracle.py).
//
// beam_wave.hip gives an utterance to one wavefront: right for a batch (four utterances per compute unit, the rest of the
// chip free for the next acoustic pass), but a lone utterance then runs on one SIMD of one CU out of 256, issuing a
// dependent instruction stream at ~10 cycles per instruction (profiles/r04_beam_sq_counters.txt): 4.3 ms at beam 100 on
// 331 frames, 91-95 % of the serving latency.  Here the SAME algorithm -- same keys, same merge arithmetic (ordered-int
// max, 2^-44 fixed-point sums: associative, hence independent of who adds first), same prune / radix select / rank rules --
// is dealt over W = 4 wavefronts, one per SIMD (profiles/r05_beam_group.txt: 4.30 -> 1.83 ms at beam 100, 2.45 -> 1.28 at 50):
//
//   * (beam, character) pair p lives in wavefront (p >> 6) % W, lane p & 63: a lane carries ceil(pairs / 256) pairs (1-3
//     instead of 5-6) through expand / score / select; beams (LM refresh, children) are dealt one per THREAD;
//   * the merge table, the beams and the radix histogram are shared in LDS (LDS atomics work across the wavefronts of a
//     workgroup); what one wavefront needs from the others crosses at LDS-only barriers:
//       B1 after the claims (all pairs sit in the table; merged slots know their contributors' maximum)
//       B4 the best combined score (prune threshold)
//       R  one per radix digit of the select (histogram buffers rotate: no clearing barrier)
//       B6 per-block (greater, equal) counts -> every wavefront computes every rank offset itself
//       Z  end of frame (children complete)
//     4-6 barriers of 4 wavefronts per general frame (50-500 cycles each, measured) against ~20 of 8 in the workgroup kernel
//     of rounds 1-3;
//   * workgroup-wide reductions (best score, live count, differing bits, claimed count) are LDS atomics on one word -- the
//     LDS serialises the lanes, one instruction per wavefront -- instead of a DPP reduction per wavefront (~25 instructions,
//     250 cycles at a lone wavefront's issue rate) plus a mailbox per wavefront;
//   * the merge table has 2 048 slots for <= 716 pairs per pass (beam_wave.hip: 512 for 358): at <= 35 % load the first probe
//     IS the compare-and-swap and hardly any lane walks on (the first version, 512 slots at 70 %, spent 7 200 of 23 400
//     cycles per frame in the probe walk of a wavefront's unluckiest lane); a frame with 100 beams x 4-7 candidates is ONE
//     pass instead of two.  A larger pass does not change the result: pairs that merge carry the same character, i.e. sit in
//     the same pass either way; the survivors of the earlier passes re-enter the last pass's prune and select, so the
//     selected SET is the top-k of the union in both forms; only the ORDER of the new beams differs (one pass: pair order;
//     several: last pass's pairs, then carried survivors; and the pair that claims a merged prefix is whoever came first).
//     Order could only break exact ties of 64-bit scores -- which almost flat posteriors do produce (tools/soak_beam.py: 3
//     cases of 2 000 differed between the kernels) -- so a tie at the cut and a tie of the final scores are decided by
//     the entries' table KEYS in both kernels, not by position: the result does not depend on the order at all;
//   * a pair that finds its key already claimed (a contributor) leaves its SCORE in the slot -- a prefix (text, last
//     character) is reached by at most four pairs: from the two beams that share its text (ending in blank / in its last
//     character) and from the two that share the text one character shorter --, and the claimer forms the log-sum-exp of all
//     of them after B1: maximum, then the terms as 2^-44 fixed-point integers, the same integers the one-wavefront kernel
//     adds atomically; its "raise the maximum", "add the terms" phases, their barrier and the 64-bit LDS atomics are gone;
//   * the first radix digit of a select starts at the first BIT in which the best score and the prune threshold differ
//     (byte-aligned digits wasted most of the first one: 3.1 -> 1.5 digits per frame at beam 100), and its histogram's total is
//     the live count: no separate count / barrier;
//   * selected pairs of a frame's LAST pass build their children straight from the registers of the lane that owns them
//     (no survivor records through LDS); frames of a blank run touch one beam per thread and skip every barrier;
//   * candidate characters depend on the frame alone: the four wavefronts list them for 32 frames at a time in parallel
//     (eight frames each, records in LDS, two barriers per 32 frames) instead of repeating every frame's list in the serial
//     loop;
//   * the final pass (commit pending words, merge identical texts, trace-back) is wavefront 0 alone, as in beam_wave.hip.
//
// Results equal beam_wave.hip's bit for bit (tests/test_beam.py::test_wave_kernel_and_group_kernel_agree,
// ::test_exact_score_ties_do_not_depend_on_the_kernel_form, every case of the randomised comparison runs both forms; soak:
// profiles/r05_beam_soak.txt, 26 000 searches, 1 / 3 / 15 rows four times each against 16 rows).  Workgroup = 256 threads = one utterance; LDS ~136 KB, one workgroup per CU.
// Used for batches of <= 64 utterances (beam_group_width below); VASR_BEAM_GROUP=0 (devtools build) pins the one-wavefront kernel.
#include <cstdlib>
#include <type_traits>

#include "beam_common.h"

namespace vasr {

namespace {
using namespace beam_detail;

constexpr int kTab = 2048;                // merge-table slots
constexpr int kFill = 716;                // pairs per pass (file header; beam_wave.hip: 358)
constexpr int kTbRows = 12;
constexpr int kLpFrames = 8;
constexpr int kLpRegs = kLpFrames * kMaxClasses / 64;
constexpr int kChars = 3072;
constexpr int kMaxBlocks = 16;            // block indices W j + w of a pass (W PPL <= 12) + 2 blocks of carried survivors
static_assert(4 * 3 + 2 <= kMaxBlocks, "block mailboxes");

template <int W>
struct GroupLds {
  unsigned long long key[2][kMaxBeams];
  unsigned long long whash[2][kMaxBeams];
  double logit[2][kMaxBeams];
  float lm_text[2][kMaxBeams];
  unsigned int meta[2][kMaxBeams];
  int ctx[2][kMaxBeams][kMaxCtx];
  float commit_lmd[2][kMaxBeams];
  int commit_wid[2][kMaxBeams];
  unsigned long long tkey[kTab];
  double tcs[kTab][3];                     // scores of the pairs that found their key already claimed (contributors: at most 3,
  int tcnt[kTab];                          //   header) and their count; the claimer merges them and resets the count
  long long sel_lgt[kMaxBeams], sel_tot[kMaxBeams];
  int sel_src[kMaxBeams];
  double fin[kMaxBeams];
  unsigned long long cmix[kMaxClasses];
  float lpq[W][kLpFrames * kMaxClasses];   // per wavefront: the log-probs of the eight frames whose candidates it lists
  // candidate records of a chunk of 8 W frames, written by the pre-pass (wavefront w lists frames 8 w .. 8 w + 7 of the chunk):
  // class ids and min(log-prob, 0) of the candidates in class order; header {count, has_space | only_blank << 1, bits of
  // min(log-prob of blank, 0), 0}
  unsigned char rc_cand[8 * W][kMaxClasses];
  float rc_val[8 * W][kMaxClasses];
  alignas(16) int rc_hdr[8 * W][4];
  alignas(16) int hist[3][256];             // radix histograms, bins in DESCENDING digit order (bin 255 - digit)
  // mailboxes
  // ... reduced by the LDS itself (one atomic per lane or per wavefront instead of a DPP reduction per wavefront -- ~25
  // instructions each at a lone wavefront's ~10 cycles per instruction -- plus a mailbox per wavefront); two sets, used by
  // alternate passes: thread 0 resets the other set right after B4
  long long red_best[2];
  int red_claimed[2];
  int mb_gt[kMaxBlocks], mb_eq[kMaxBlocks];
  int merge_epoch, anychar_epoch, overflow;
  int n_log;
};
static_assert(2 * kTbRows * kMaxBeams * 4 <= (int)(sizeof(unsigned long long) * kTab), "trace-back batches alias tkey");
static_assert(kChars * 2 <= (int)(sizeof(unsigned long long) * 2 * kMaxBeams * 3), "transcript characters alias the beam keys / hashes / logits");
static_assert(sizeof(GroupLds<4>) <= 160 * 1024, "one workgroup per compute unit");
static_assert(kFill <= 3 * 256 && kFill <= kTab * 7 / 20, "pairs per pass: three per lane at most, table at most 35 % full");

constexpr unsigned kMetaCached = 1u << 24, kMetaCommit = 1u << 25, kMetaOov = 1u << 26;   // (beam_wave.hip WaveLds::meta)
constexpr int kSrcOov = 1 << 16;          // a pair record (beam << 8 | class) carries its child's "OOV" bit here
__device__ inline int meta_last(unsigned m) { return (int)(m & 0xffu) - 1; }
__device__ inline int meta_wlen(unsigned m) { return (int)((m >> 8) & 0xffffu); }
__device__ inline unsigned make_meta(int last, int wlen, unsigned flags) {
  return (unsigned)(last + 1) | ((unsigned)min(wlen, 0xffff) << 8) | flags;
}

__device__ inline void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// LDS-only workgroup barrier: __syncthreads() would also drain the vector-memory counter, i.e. wait for the back-pointer
// stores and the prefetched log-probs at every phase boundary
__device__ inline void group_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// (unsigned long long)((double)e * 2^44) for a float e in [0, 1] -- the 2^-44 fixed-point term of a merge -- in integer
// arithmetic: e = m * 2^(E - 150), so the product is m shifted by E - 106, truncated.  The double form is ~15 instructions of
// conversion per term on this ISA (there is no f64 -> u64 instruction).  Identical for EVERY float in [0, 1]: checked
// exhaustively on the host (1 056 964 610 values; tests/test_beam.py::test_fixed_point_term_integer_form samples it).
__device__ inline unsigned long long fix44(float e) {
  const unsigned b = __float_as_uint(e);
  const int E = (int)((b >> 23) & 0xffu);
  const unsigned long long m = (unsigned long long)((b & 0x7fffffu) | 0x800000u);
  const int sh = E - 106;
  return E == 0 ? 0ull : (sh >= 0 ? m << sh : (sh > -64 ? m >> -sh : 0ull));
}

__device__ inline int lane_id() { return (int)(threadIdx.x & 63); }
__device__ inline int rank_in(unsigned long long mask) {
  return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
}

// grid (B), block 64 W: workgroup g searches utterance g
template <int W>
__global__ __launch_bounds__(64 * W) void beam_group_kernel(const float* __restrict__ logp, int batch, int frames_ld,
                                                             const int32_t* __restrict__ row_frames, int V1, int space_id,
                                                             int beam_width, float token_min_logp, float beam_prune_logp,
                                                             LmView lm, int use_lm, unsigned int* __restrict__ bp_all,
                                                             unsigned long long* __restrict__ eoslog_all,
                                                             int32_t* __restrict__ out_ids, int32_t* __restrict__ out_len,
                                                             float* __restrict__ out_score) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  using Lds = GroupLds<W>;
  Lds& S = *reinterpret_cast<Lds*>(smem);
  const int tid = (int)threadIdx.x, lane = lane_id();
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = (int)blockIdx.x;
  const int V = V1 - 1;
  const bool trie = use_lm && lm.trie != nullptr;             // pyctcdecode's unigram set + character trie (".arpa" semantics)
  const int frames = row_frames ? max(0, min(frames_ld, row_frames[b])) : frames_ld;
  const float* lrow = logp + (int64_t)b * frames_ld * V1;
  unsigned int* bp = bp_all + (int64_t)b * frames_ld * kMaxBeams;
  unsigned long long* eoslog = eoslog_all + (int64_t)b * frames_ld * kMaxBeams;

  for (int i = tid; i < kTab; i += 64 * W) { S.tkey[i] = 0; S.tcnt[i] = 0; }
  for (int c = tid; c < kMaxClasses; c += 64 * W) S.cmix[c] = hmix(hmix(kFnvOffset, (unsigned long long)(c + 7)), 0x9e3779b9ull);
  for (int i = tid; i < 3 * 256; i += 64 * W) (&S.hist[0][0])[i] = 0;
  if (tid == 0) {
    S.key[0][0] = kFnvOffset; S.whash[0][0] = kFnvOffset; S.logit[0][0] = 0.0; S.lm_text[0][0] = 0.f;
    S.meta[0][0] = make_meta(-1, 0, 0);
    for (int i = 0; i < kMaxCtx; ++i) S.ctx[0][0][i] = -1;
    if (use_lm) S.ctx[0][0][kMaxCtx - 1] = lm.bos;
    S.commit_lmd[0][0] = 0.f; S.commit_wid[0][0] = 0;
    S.merge_epoch = -1; S.anychar_epoch = -1; S.n_log = 0; S.overflow = 0;
    for (int k = 0; k < 2; ++k) { S.red_best[k] = ord64(-1e300); S.red_claimed[k] = 0; }
  }
  group_sync();
  int cur = 0, nb = 1, epoch = 0;
  int hd = 0;                  // radix digits histogrammed so far: digit hd uses hist[hd % 3] (uniform)
  bool all_blank = false;      // every live beam ends in blank (uniform)
  bool dirty = false;          // frames of a blank run have updated beams without a barrier (uniform)

  // Candidate characters are a function of the frame alone, not of the beams: the four wavefronts list them for 32 frames
  // at a time IN PARALLEL (eight frames each) instead of each repeating every frame's list in the serial loop -- 1 250 of a
  // frame's 11 300 cycles at beam 50 were "loop top + candidates".  A wavefront stages the log-probs of ITS eight frames
  // (requested one chunk ahead, unconditional clamped loads: beam_wave.hip says why) and writes the records to LDS.
  constexpr int kChunk = 8 * W;
  float q[kLpRegs];
  const int lp_batch = kLpFrames * V1;
  auto lp_request = [&](int t0) __attribute__((always_inline)) {
    const int n = min(lp_batch, (frames - t0) * V1);
    if (n <= 0) return;
    const float* src = lrow + (int64_t)t0 * V1;
#pragma unroll
    for (int k = 0; k < kLpRegs; ++k) q[k] = src[min(64 * k + lane, n - 1)];
  };
  lp_request(kLpFrames * wv);
  float* lpq = S.lpq[wv];

  // One new beam at rank r from pair (parent bi, character c) with merged logit bits lgt
  auto build_child = [&](int t, int r, int src, long long lgt, bool has_space) __attribute__((always_inline)) {
    const int nxt = cur ^ 1;
    const int bi = (src >> 8) & 255, c = src & 255;
    const unsigned m = S.meta[cur][bi];
    const int last = meta_last(m), wlen = meta_wlen(m);
    const bool stay = (c == V || c == last);
    unsigned long long key = S.key[cur][bi], whash = S.whash[cur][bi];
    float lm_text = S.lm_text[cur][bi];
    const int4 ctx_p = *reinterpret_cast<const int4*>(&S.ctx[cur][bi][0]);
    int4 ctx_n = ctx_p;
    const float p_lmd = S.commit_lmd[cur][bi];
    const int p_wid = S.commit_wid[cur][bi];
    int wlen_new = wlen;
    unsigned int appended = 0;
    unsigned flags = (src & kSrcOov) ? kMetaOov : 0u;       // (the score step decided it: same pending word, same bit)
    if (stay) {
      if ((m & kMetaCached) || (has_space && wlen > 0)) flags |= kMetaCached;
      flags |= m & kMetaCommit;
    } else if (c == space_id) {
      if (wlen > 0) {
        key = hmix(key, (unsigned long long)c);
        appended = c + 1;
        if (use_lm) {
          lm_text += p_lmd;
          ctx_n = make_int4(ctx_p.y, ctx_p.z, ctx_p.w, p_wid);
        }
        wlen_new = 0; whash = kFnvOffset;
      }
    } else {
      key = hmix(key, (unsigned long long)c);
      whash = hmix(whash, (unsigned long long)c);
      wlen_new = wlen + 1;
      appended = c + 1;
    }
    if (c != V) S.anychar_epoch = t;                 // (every writer stores the same value)
    S.key[nxt][r] = key; S.whash[nxt][r] = whash;
    S.logit[nxt][r] = __longlong_as_double(lgt);
    S.lm_text[nxt][r] = lm_text;
    S.meta[nxt][r] = make_meta(c, wlen_new, flags);
    *reinterpret_cast<int4*>(&S.ctx[nxt][r][0]) = ctx_n;
    S.commit_lmd[nxt][r] = p_lmd;
    S.commit_wid[nxt][r] = p_wid;
    bp[(int64_t)t * kMaxBeams + r] = ((unsigned)bi << 8) | appended;
  };

  for (int t = 0; t < frames; ++t) {
    // ---- 1. candidate characters: from the chunk's records (pre-pass at every chunk boundary) ----
    if ((t & (kChunk - 1)) == 0) {
      group_sync();                        // the previous chunk's records (and frame t - 1's beams) are done with
      dirty = false;
#pragma unroll
      for (int k = 0; k < kLpRegs; ++k) if (64 * k + lane < lp_batch) lpq[64 * k + lane] = q[k];
      lp_request(t + kChunk + kLpFrames * wv);
      wave_sync();
      const int c0 = lane, c1 = lane + 64;
#pragma unroll 1
      for (int f = 0; f < kLpFrames; ++f) {
        const int tt = t + kLpFrames * wv + f;
        if (tt >= frames) break;
        const int fr = tt & (kChunk - 1);
        const float* lq = lpq + f * V1;
        const float x0 = c0 < V1 ? lq[c0] : 0.f, x1 = c1 < V1 ? lq[c1] : 0.f;
        // pyctcdecode works on log(clip(p, 1e-15, 1)) = clip(x, log 1e-15, 0); for a class that can be a candidate that is min(x, 0)
        const float v0 = fminf(fmaxf(x0, -34.538776f), 0.f), v1 = fminf(fmaxf(x1, -34.538776f), 0.f);
        auto okey = [](float v) { const unsigned q = __float_as_uint(v); return (q & 0x80000000u) ? ~q : (q | 0x80000000u); };
        // candidates = {x >= token_min_logp} U {arg-max}: when any class passes the threshold the arg-max is among them already
        bool k0 = c0 < V1 && v0 >= token_min_logp, k1 = c1 < V1 && v1 >= token_min_logp;
        unsigned long long m0 = __ballot(k0), m1 = __ballot(k1);
        if ((m0 | m1) == 0ull) {
          const unsigned key0 = c0 < V1 ? okey(v0) : 0u, key1 = c1 < V1 ? okey(v1) : 0u;
          const unsigned kmax = wave_max_u32(max(key0, key1));
          const unsigned long long a0 = __ballot(c0 < V1 && key0 == kmax), a1 = __ballot(c1 < V1 && key1 == kmax);
          const int amax = a0 ? __ffsll((long long)a0) - 1 : 64 + __ffsll((long long)a1) - 1;   // first maximum
          k0 = c0 == amax; k1 = c1 == amax;
          m0 = __ballot(k0); m1 = __ballot(k1);
        }
        if (k0) { const int r = rank_in(m0); S.rc_cand[fr][r] = (unsigned char)c0; S.rc_val[fr][r] = fminf(x0, 0.f); }
        if (k1) { const int r = __popcll(m0) + rank_in(m1); S.rc_cand[fr][r] = (unsigned char)c1; S.rc_val[fr][r] = fminf(x1, 0.f); }
        const int n_c = __popcll(m0) + __popcll(m1);
        const bool sp = space_id < 64 ? (m0 >> space_id & 1) : (space_id < 128 ? (m1 >> (space_id - 64) & 1) : false);
        const bool ob = n_c == 1 && (V < 64 ? (m0 >> V & 1) : (m1 >> (V - 64) & 1));
        if (lane == 0) {
          S.rc_hdr[fr][0] = n_c; S.rc_hdr[fr][1] = (sp ? 1 : 0) | (ob ? 2 : 0);
          S.rc_hdr[fr][2] = __float_as_int(fminf(lq[V], 0.f));
        }
      }
      group_sync();
    }
    const int fr = t & (kChunk - 1);
    const unsigned char* cand = S.rc_cand[fr];
    const float* cval = S.rc_val[fr];
    const int4 hdr = *reinterpret_cast<const int4*>(&S.rc_hdr[fr][0]);
    const int nc_all = __builtin_amdgcn_readfirstlane(hdr.x);
    const bool has_space = __builtin_amdgcn_readfirstlane(hdr.y) & 1, only_blank = __builtin_amdgcn_readfirstlane(hdr.y) & 2;

    // a blank-only frame met by beams that all end in blank: one beam per thread, nothing crosses wavefronts
    if (only_blank && all_blank) {
      const double add = (double)__int_as_float(__builtin_amdgcn_readfirstlane(hdr.z));
      for (int i = tid; i < nb; i += 64 * W) {
        S.logit[cur][i] += add;
        bp[(int64_t)t * kMaxBeams + i] = (unsigned)i << 8;
      }
      dirty = true;
      continue;
    }
    if (dirty) { group_sync(); dirty = false; }

    // ---- 2. ' ' is a candidate: LM cache log + commit scores, one beam per thread ----
    if (use_lm && has_space) {
      for (int i0 = 0; i0 < nb; i0 += 64 * W) {
        const int i = i0 + tid;
        bool put = false;
        unsigned long long h = 0;
        if (i < nb) {
          const unsigned m = S.meta[cur][i];
          if (meta_wlen(m) > 0) {
            put = !(m & kMetaCached);
            h = hmix(S.key[cur][i], (unsigned long long)space_id) | 1ull;
            if (!(m & kMetaCommit)) {
              int ctx[kMaxCtx];
#pragma unroll
              for (int qq = 0; qq < kMaxCtx; ++qq) ctx[qq] = S.ctx[cur][i][qq];
              int w;
              S.commit_lmd[cur][i] = lm_word_score(lm, ctx, S.whash[cur][i], false, &w);
              S.commit_wid[cur][i] = w;
              S.meta[cur][i] = m | kMetaCommit;
            }
          }
        }
        const unsigned long long pm = __ballot(put);
        if (pm) {                                    // the log is a SET of keys: its order does not matter
          int base = 0;
          if (lane == 0) base = atomicAdd(&S.n_log, __popcll(pm));
          base = __builtin_amdgcn_readfirstlane(base);
          if (put) eoslog[base + rank_in(pm)] = h;
        }
      }
      // (the commit scores are read after barrier B1)
    }

    const int cap = max(1, (int)(((float)kFill + 0.5f) * __builtin_amdgcn_rcpf((float)nb)));
    // survivors carried from the earlier passes of this frame (wavefront 0 only): ranks lane and lane + 64
    long long c_tot[2] = {ord64(-1e300), ord64(-1e300)}, c_lgt[2] = {0, 0};
    int c_src[2] = {0, 0};
    int n_sel = 0;

    // table key of pair (beam bi, character c) = src -- (prefix text, last character), as the expand step forms it
    auto pair_key = [&](int sr) __attribute__((always_inline)) -> unsigned long long {
      const int bi = (sr >> 8) & 255, c = sr & 255;
      const unsigned m = S.meta[cur][bi];
      const bool grows = !(c == V || c == meta_last(m)) && !(c == space_id && meta_wlen(m) == 0);
      const unsigned long long key = S.key[cur][bi];
      return ((grows ? hmix(key, (unsigned long long)c) : key) ^ S.cmix[c]) | 1ull;
    };

    // carry_tag: the frame's earlier passes left survivors (wavefront 0 carries them as two more blocks of entries); a
    // single-pass frame -- the usual case -- compiles them out
    auto pass = [&](auto ppl_tag, auto carry_tag, int c_lo, int nc, bool last_pass) __attribute__((always_inline)) {
      constexpr int PPL = decltype(ppl_tag)::value;
      constexpr int NC = decltype(carry_tag)::value ? 2 : 0;
      const int npairs = nb * nc;
      const float inv_nc = __builtin_amdgcn_rcpf((float)nc);
      ++epoch;
      int slot[PPL], src[PPL];
      double score[PPL];
      unsigned claimed = 0, act = 0;
      unsigned long long kk[PPL];
      int stride[PPL];
      // character trie (LM built with a unigram list): the home bucket of the child's pending word is requested here, the
      // score step -- behind the claims and barrier B1 -- reads the answer
      ulonglong2 tfirst[PPL];
      unsigned long long wnew[PPL];
#pragma unroll
      for (int j = 0; j < PPL; ++j) {
        const int p = 64 * (W * j + wv) + lane;
        if (p < npairs) act |= 1u << j;
        const int pp = min(p, npairs - 1);
        const int bi = (int)(((float)pp + 0.5f) * inv_nc);
        const int ci = c_lo + pp - bi * nc;
        const int c = cand[ci];
        const unsigned m = S.meta[cur][bi];
        const int last = meta_last(m);
        unsigned long long key = S.key[cur][bi];
        score[j] = S.logit[cur][bi] + (double)cval[ci];
        src[j] = (bi << 8) | c;
        const bool grows = !(c == V || c == last) && !(c == space_id && meta_wlen(m) == 0);
        const unsigned long long kx = hmix(key, (unsigned long long)c);
        key = grows ? kx : key;
        const unsigned long long k = (key ^ S.cmix[c]) | 1ull;
        kk[j] = k;
        slot[j] = (int)((k >> 17) & (kTab - 1));
        stride[j] = (int)((k >> 40) & (kTab - 1)) | 1;
        if (trie) {                                                                     // (uniform)
          wnew[j] = hmix(S.whash[cur][bi], (unsigned long long)c);
          tfirst[j] = trie_first(lm, wnew[j]);
        }
      }
      // the table is at most a sixth full: a home slot is usually empty, so the first probe IS the compare-and-swap (one LDS
      // round trip instead of read + swap); its answer says empty (claimed), our key (a merge) or a foreign key (walk on)
      unsigned long long seen[PPL];
#pragma unroll
      for (int j = 0; j < PPL; ++j)
        if (act >> j & 1) {
          seen[j] = atomicCAS(&S.tkey[slot[j]], 0ull, kk[j]);
          if (seen[j] == 0ull) { claimed |= 1u << j; seen[j] = kk[j]; }
        }
#pragma unroll
      for (int j = 0; j < PPL; ++j) {
        if (act >> j & 1) {
          int i = slot[j];
          const unsigned long long k = kk[j];
          if (seen[j] != k) {
            while (true) {
              i = (i + stride[j]) & (kTab - 1);
              const unsigned long long old = atomicCAS(&S.tkey[i], 0ull, k);
              if (old == 0ull) { claimed |= 1u << j; break; }
              if (old == k) break;
            }
          }
          if (!(claimed >> j & 1)) {                                   // a contributor: its score goes into the slot's next cell
            const int n = atomicAdd(&S.tcnt[i], 1);
            if (n < 3) S.tcs[i][n] = score[j];
            else S.overflow = 1;                                       // (cannot happen -- a prefix has at most four pairs: file header -- and is reported: id_len = -1)
          }
          slot[j] = i;
        }
      }
      if (__ballot((act & ~claimed) != 0u) != 0ull && lane == 0) S.merge_epoch = epoch;
      {
        int n_cl = 0;
#pragma unroll
        for (int j = 0; j < PPL; ++j) n_cl += __popcll(__ballot(claimed >> j & 1));
        if (lane == 0 && n_cl) atomicAdd(&S.red_claimed[epoch & 1], n_cl);
      }
      group_sync();                                                     // ---- B1
       
      const bool any_merge = S.merge_epoch == epoch;                    // uniform over the workgroup
      // ---- 3. merged prefixes, each in the lane that claimed its slot ----
      long long tot[PPL], lgt[PPL];
      long long my_best = NC ? max(c_tot[0], c_tot[1]) : ord64(-1e300);
#pragma unroll
      for (int j = 0; j < PPL; ++j) {
        const bool mine = claimed >> j & 1;
        const int i = slot[j], bi = src[j] >> 8, c = src[j] & 255;
        float lmt = 0.f;
        if (use_lm) {
          const unsigned m = S.meta[cur][bi];
          const int last = meta_last(m), wlen = meta_wlen(m);
          const bool stay = (c == V || c == last);
          const int wlen_new = stay ? wlen : (c == space_id ? 0 : wlen + 1);
          const float commit = S.commit_lmd[cur][bi];
          // partial_penalty(unk_offset, wlen_new), its division only when some lane's pending word is longer than six
          // characters (as a select the compiler runs the ~12-instruction division in every frame)
          // is_oov of the child's pending word: the parent's when the word stays; once outside the trie, outside for good
          bool oov = true;
          if (trie) {
            if (stay) oov = (m & kMetaOov) != 0u;
            else if (c != space_id && !(wlen > 0 && (m & kMetaOov))) oov = !trie_has_node(lm, wnew[j], tfirst[j]);
          }
          if (oov) src[j] |= kSrcOov;
          float pen = (wlen_new > 0 && oov) ? lm.unk_offset : 0.f;
          if (__ballot(wlen_new > 6) != 0ull) pen = wlen_new > 6 ? pen * (float)wlen_new / 6.0f : pen;
          lmt = S.lm_text[cur][bi] + pen + ((!stay && c == space_id && wlen > 0) ? commit : 0.f);
        }
        double logit = score[j];
        if (mine) {
          S.tkey[i] = 0;
          if (any_merge) {
            const int n = S.tcnt[i];
            if (n > 0) {
              // log-sum-exp of the claimer and its n contributors: the maximum m first, then every term exp(s - m) through the
              // hardware 2^x on a float (1 ulp) as a 2^-44 fixed-point integer -- integer sums are associative, so the result
              // does not depend on who arrived first (and equals the one-wavefront kernel's max / atomic-add form bit for bit)
              S.tcnt[i] = 0;
              // (a second or third contributor is rare: their cells are read and their terms formed only when some lane has one)
              const bool any2 = __ballot(n >= 2) != 0ull, any3 = __ballot(n >= 3) != 0ull;
              const double s0 = S.tcs[i][0];
              const double s1 = any2 ? S.tcs[i][1] : s0, s2 = any3 ? S.tcs[i][2] : s0;
              double m = fmax(score[j], s0);
              if (any2) m = n >= 2 ? fmax(m, s1) : m;
              if (any3) m = n >= 3 ? fmax(m, s2) : m;
              unsigned long long s8 = fix44(__builtin_amdgcn_exp2f((float)((score[j] - m) * 1.4426950408889634)));
              s8 += fix44(__builtin_amdgcn_exp2f((float)((s0 - m) * 1.4426950408889634)));
              if (any2 && n >= 2) s8 += fix44(__builtin_amdgcn_exp2f((float)((s1 - m) * 1.4426950408889634)));
              if (any3 && n >= 3) s8 += fix44(__builtin_amdgcn_exp2f((float)((s2 - m) * 1.4426950408889634)));
              logit = m + (s8 == (unsigned long long)kFix ? 0.0 : log_ge1((double)s8 * (1.0 / kFix)));
            }
          }
        }
        tot[j] = mine ? ord64(logit + (double)lmt) : ord64(-1e300);
        lgt[j] = __double_as_longlong(logit);
        my_best = max(my_best, tot[j]);
      }
      if (my_best != ord64(-1e300)) atomicMax(&S.red_best[epoch & 1], my_best);
      group_sync();                                                     // ---- B4
      const long long best = S.red_best[epoch & 1];
      const int n_claimed_all = S.red_claimed[epoch & 1] + n_sel;       // + the carried survivors
      if (tid == 0) {   // the other set: last read a pass ago, next written after this pass's B6
        const int o = (epoch + 1) & 1;
        S.red_best[o] = ord64(-1e300); S.red_claimed[o] = 0;
      }
      // ---- 4. prune (max + beam_prune_logp), then the top beam_width by combined score ----
      const long long thr_prune = ord64(unord64(best) + (double)beam_prune_logp);
      const unsigned long long ubest = (unsigned long long)best ^ 0x8000000000000000ull;
      unsigned live = 0;
#pragma unroll
      for (int j = 0; j < PPL + NC; ++j) {
        const long long tt = j < PPL ? tot[j] : c_tot[j - PPL];
        const bool lv = j < PPL ? ((claimed >> j & 1) && tt >= thr_prune) : (wv == 0 && lane + 64 * (j - PPL) < n_sel && tt >= thr_prune);
        if (lv) live |= 1u << j;
      }
      unsigned long long prefix = 0, mask = 0;
      int want = beam_width;
      bool tie = false;                                                 // (uniform) the digits ran out on a bucket with more entries than wanted
      if (n_claimed_all > beam_width) {                                 // (uniform over the workgroup) only then can a select be needed
        // Every live key lies between the prune threshold and the best score, so the leading bits those two have in common
        // (sign, exponent, the top of the mantissa) are common to all of them: the first digit starts at the first bit in
        // which they differ -- known to every wavefront from `best` alone.  (An earlier version counted the live entries and
        // OR-ed their differing bits over the workgroup first: one more barrier and two atomics per select frame, for a
        // first digit that started at most a bit or two lower.)  The first histogram's total IS the live count: if it does
        // not exceed beam_width nothing is selected away.  Digits then walk down in steps of eight, the last one clamped to
        // bits 7..0 (an overlap with known bits is harmless: they match).
        const unsigned long long ulo = (unsigned long long)thr_prune ^ 0x8000000000000000ull;
        const unsigned long long d0 = ubest ^ ulo;
        const int lead = d0 ? __clzll((long long)d0) : 64;
        if (lead > 0) { mask = lead == 64 ? ~0ull : (~0ull << (64 - lead)); prefix = ubest & mask; }
        tie = lead == 64;                                               // (a prune threshold AT the best score: whatever is live is tied)
        bool first = true;
        // Histogram buffers rotate with a running digit count: digit hd adds into hist[hd % 3] -- cleared during digit
        // hd - 1 (or at the start) -- and clears hist[(hd + 1) % 3], last READ during digit hd - 2, which every wavefront
        // has left behind when it passed the barrier of digit hd - 1.  No clearing barrier.
#pragma unroll 1
        for (int shift = max(0, 56 - lead); lead < 64; shift = max(0, shift - 8)) {
          int* h = S.hist[hd % 3];
          int* hn = S.hist[(hd + 1) % 3];
          ++hd;
#pragma unroll
          for (int j = 0; j < PPL + NC; ++j) {
            const unsigned long long u = (unsigned long long)(j < PPL ? tot[j] : c_tot[j - PPL]) ^ 0x8000000000000000ull;
            if ((live >> j & 1) && (u & mask) == prefix) atomicAdd(&h[255 - (int)((u >> shift) & 255)], 1);
          }
          for (int i = tid; i < 256; i += 64 * W) hn[i] = 0;          // the next digit's buffer (last read two digits ago)
          group_sync();                                               // ---- R
          
          // lane l owns digits 255 - 4 l ... 252 - 4 l = bins 4 l ... 4 l + 3: one 16-byte read, the largest digit first
          const int4 c4 = *reinterpret_cast<const int4*>(&h[4 * lane]);
          const int cnt[4] = {c4.x, c4.y, c4.z, c4.w};
          const int mine = (c4.x + c4.y) + (c4.z + c4.w);
          const int incl = wave_scan_incl(mine);
          if (first) {
            first = false;
            if (__builtin_amdgcn_readlane(incl, 63) <= want) { mask = 0; prefix = 0; break; }   // no more live entries than beams
          }
          int above = incl - mine;
          int f_bucket = -1, f_want = 0, f_whole = 0;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if (above < want && want <= above + cnt[j]) { f_bucket = 255 - (4 * lane + j); f_want = want - above; f_whole = cnt[j] == want - above; }
            above += cnt[j];
          }
          const unsigned long long fm = __ballot(f_bucket >= 0);
          const int fl = __ffsll((long long)fm) - 1;
          const int bucket = __builtin_amdgcn_readlane(f_bucket, fl);
          want = __builtin_amdgcn_readlane(f_want, fl);
          const int whole = __builtin_amdgcn_readlane(f_whole, fl);
          prefix |= (unsigned long long)bucket << shift;
          mask |= 0xFFull << shift;
          if (whole || shift == 0) { tie = !whole; break; }
        }
        // ---- an exact tie at the cut: more entries share all 64 bits of the cut score than fit.  The ORDER of the entries is
        //      not the same in the two kernels (passes of 716 pairs here, 358 there; here the pair that claims a merged prefix
        //      is whoever came first), so position must not decide who stays: the `want` tied entries with the LARGEST table
        //      keys do -- (prefix text, last character), unique per entry and the same in every schedule.  The same digit
        //      search over the keys of the tied entries (one digit as a rule); the others then leave the live set, and what
        //      follows sees exactly `want` entries equal to the cut.  Out of line: flat synthetic posteriors reach this
        //      (tools/soak_beam.py), a model's do not ----
        if (tie) {
          const int want_tied = want;
          unsigned long long tk[PPL + NC], kprefix = 0, kmask = 0;
          unsigned tied = 0;
#pragma unroll
          for (int j = 0; j < PPL + NC; ++j) {
            tk[j] = pair_key(j < PPL ? src[j] : c_src[j - PPL]);
            const unsigned long long u = (unsigned long long)(j < PPL ? tot[j] : c_tot[j - PPL]) ^ 0x8000000000000000ull;
            if ((live >> j & 1) && u == prefix) tied |= 1u << j;
          }
          bool fits = false;
#pragma unroll 1
          for (int shift = 56;; shift -= 8) {
            int* h = S.hist[hd % 3];
            int* hn = S.hist[(hd + 1) % 3];
            ++hd;
#pragma unroll
            for (int j = 0; j < PPL + NC; ++j)
              if ((tied >> j & 1) && (tk[j] & kmask) == kprefix) atomicAdd(&h[255 - (int)((tk[j] >> shift) & 255)], 1);
            for (int i = tid; i < 256; i += 64 * W) hn[i] = 0;
            group_sync();                                               // ---- R (tie)
            const int4 c4 = *reinterpret_cast<const int4*>(&h[4 * lane]);
            const int cnt[4] = {c4.x, c4.y, c4.z, c4.w};
            const int mine = (c4.x + c4.y) + (c4.z + c4.w);
            const int incl = wave_scan_incl(mine);
            if (shift == 56 && __builtin_amdgcn_readlane(incl, 63) <= want) { fits = true; break; }   // (lead == 64 only) all tied entries fit
            int above = incl - mine;
            int f_bucket = -1, f_want = 0, f_whole = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              if (above < want && want <= above + cnt[j]) { f_bucket = 255 - (4 * lane + j); f_want = want - above; f_whole = cnt[j] == want - above; }
              above += cnt[j];
            }
            const unsigned long long fm = __ballot(f_bucket >= 0);
            const int fl = __ffsll((long long)fm) - 1;
            const int bucket = __builtin_amdgcn_readlane(f_bucket, fl);
            want = __builtin_amdgcn_readlane(f_want, fl);
            const int whole = __builtin_amdgcn_readlane(f_whole, fl);
            kprefix |= (unsigned long long)bucket << shift;
            kmask |= 0xFFull << shift;
            if (whole || shift == 0) break;
          }
          if (!fits) {
#pragma unroll
            for (int j = 0; j < PPL + NC; ++j)
              if ((tied >> j & 1) && (tk[j] & kmask) < kprefix) live &= ~(1u << j);
          }
          want = want_tied;
        }
      }
      // ---- selected: live and key > threshold prefix, plus the first `want` equal to it in PAIR ORDER (block = 64 pairs:
      //      index W j + w; the carried survivors follow as blocks W PPL and W PPL + 1).  Every wavefront publishes its
      //      blocks' (greater, equal) counts; after B6 each computes every block's offset itself ----
      bool gt[PPL + NC], eq[PPL + NC];
#pragma unroll
      for (int j = 0; j < PPL + NC; ++j) {
        const long long tt = j < PPL ? tot[j] : c_tot[j - PPL];
        gt[j] = false; eq[j] = false;
        if (live >> j & 1) {
          const unsigned long long u = ((unsigned long long)tt ^ 0x8000000000000000ull) & mask;
          if (mask == 0 || u > prefix) gt[j] = true; else if (u == prefix) eq[j] = true;
        }
        const int ng = __popcll(__ballot(gt[j])), ne = __popcll(__ballot(eq[j]));
        if (lane == 0) {
          if (j < PPL) { S.mb_gt[W * j + wv] = ng; S.mb_eq[W * j + wv] = ne; }
          else if (wv == 0) { S.mb_gt[W * PPL + (j - PPL)] = ng; S.mb_eq[W * PPL + (j - PPL)] = ne; }
        }
      }
      group_sync();                                                     // ---- B6
      // (every wavefront walks all blocks' counts itself; a lane-per-block form with two DPP scans measured 160-350 cycles per
      // frame SLOWER: at one to three pairs per lane the walk is 4-14 short iterations)
      int n_out = 0, eq_seen = 0;
      int off_out[PPL + NC], off_eq[PPL + NC];
#pragma unroll
      for (int blk = 0; blk < W * PPL + NC; ++blk) {
        const int ng = S.mb_gt[blk], ne = S.mb_eq[blk];
        const int j = blk < W * PPL ? blk / W : PPL + (blk - W * PPL);
        const bool own = blk < W * PPL ? (blk % W == wv) : (wv == 0);
        if (own) { off_out[j] = n_out; off_eq[j] = eq_seen; }
        n_out += ng + min(ne, max(0, want - eq_seen));
        eq_seen += ne;
      }
#pragma unroll
      for (int j = 0; j < PPL + NC; ++j) {
        if (j >= PPL && wv != 0) continue;
        const unsigned long long em = __ballot(eq[j]);
        const bool take = gt[j] || (eq[j] && off_eq[j] + rank_in(em) < want);
        const unsigned long long tm = __ballot(take);
        const int dst = off_out[j] + rank_in(tm);
        if (take && dst < kMaxBeams) {
          const int sr = j < PPL ? src[j] : c_src[j - PPL];
          const long long lg = j < PPL ? lgt[j] : c_lgt[j - PPL];
          if (last_pass) build_child(t, dst, sr, lg, has_space);
          else { S.sel_src[dst] = sr; S.sel_lgt[dst] = lg; S.sel_tot[dst] = j < PPL ? tot[j] : c_tot[j - PPL]; }
        }
      }
      n_sel = min(n_out, kMaxBeams);
      if (!last_pass) {
        group_sync();                                                   // ---- B7 (multi-pass frames only)
        if (wv == 0) {
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const int r = lane + 64 * j;
            if (r < n_sel) { c_tot[j] = S.sel_tot[r]; c_lgt[j] = S.sel_lgt[r]; c_src[j] = S.sel_src[r]; }
            else { c_tot[j] = ord64(-1e300); c_lgt[j] = 0; c_src[j] = 0; }
          }
        }
        // (the records are rewritten after the next pass's B6 only: wavefront 0 has long read them by then)
      }
    };

#pragma unroll 1
    for (int c_lo = 0; c_lo < nc_all; c_lo += cap) {
      const int nc = min(cap, nc_all - c_lo);
      const bool last_pass = c_lo + cap >= nc_all;
      const int npairs = nb * nc;
      const int ppl = (npairs + 64 * W - 1) / (64 * W);
      auto go = [&](auto ppl_tag) __attribute__((always_inline)) {
        if (n_sel > 0) pass(ppl_tag, std::true_type{}, c_lo, nc, last_pass);
        else pass(ppl_tag, std::false_type{}, c_lo, nc, last_pass);
      };
      if (ppl <= 1) go(std::integral_constant<int, 1>{});
      else if (ppl == 2) go(std::integral_constant<int, 2>{});
      else go(std::integral_constant<int, 3>{});
    }
    group_sync();                                                       // ---- Z: the new beams are complete
    all_blank = S.anychar_epoch != t;
    nb = n_sel;
    cur ^= 1;
  }
  // every wavefront's back-pointer and log stores have reached L2 before wavefront 0 reads them back
  __syncthreads();
  if (wv != 0) return;
  const int n_log = S.n_log;

  // ---- final: commit pending words (LM score with </s>), merge identical texts, pick the best (as beam_wave.hip) ----
  int in_cache[2] = {0, 0};
  if (use_lm) {
    int myslot[2] = {-1, -1};
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int i = lane + 64 * j;
      if (i < nb) {
        const unsigned m = S.meta[cur][i];
        if (meta_wlen(m) > 0) {
          in_cache[j] = (m & kMetaCached) ? 1 : 0;
          if (!in_cache[j] && n_log > 0) {
            const unsigned long long k = hmix(S.key[cur][i], (unsigned long long)space_id) | 1ull;
            int q2 = (int)((k >> 17) & (kTab - 1));
            while (true) {
              const unsigned long long old = atomicCAS(&S.tkey[q2], 0ull, k);
              if (old == 0ull || old == k) break;
              q2 = (q2 + 1) & (kTab - 1);
            }
            myslot[j] = q2;
          }
        }
      }
    }
    for (int i = lane; i < kTab; i += 64) S.tcnt[i] = 0;
    wave_sync();
    for (int q2 = lane; q2 < n_log; q2 += 64) {
      const unsigned long long k = __hip_atomic_load(&eoslog[q2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      for (int i = (int)((k >> 17) & (kTab - 1));; i = (i + 1) & (kTab - 1)) {
        const unsigned long long e = S.tkey[i];
        if (e == k) { S.tcnt[i] = 1; break; }
        if (e == 0) break;
      }
    }
    wave_sync();
#pragma unroll
    for (int j = 0; j < 2; ++j) if (myslot[j] >= 0) in_cache[j] = S.tcnt[myslot[j]];
    wave_sync();
  }
  double* fin = S.fin;
  unsigned long long* fkey = reinterpret_cast<unsigned long long*>(S.sel_lgt);
  double* frank = reinterpret_cast<double*>(S.sel_tot);
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int i = lane + 64 * j;
    if (i < nb) {
      const unsigned m = S.meta[cur][i];
      const int wlen = meta_wlen(m);
      double total = S.logit[cur][i];
      if (use_lm) {
        float lmv = S.lm_text[cur][i];
        if (wlen > 0) {
          int ctx[kMaxCtx], wid;
#pragma unroll
          for (int q2 = 0; q2 < kMaxCtx; ++q2) ctx[q2] = S.ctx[cur][i][q2];
          lmv += lm_word_score(lm, ctx, S.whash[cur][i], !in_cache[j], &wid);
        }
        total += (double)lmv;
      }
      fin[i] = total;
      fkey[i] = wlen > 0 ? hmix(S.key[cur][i], (unsigned long long)space_id) : S.key[cur][i];
      frank[i] = S.logit[cur][i] + (use_lm ? (double)(S.lm_text[cur][i] + partial_penalty(lm.unk_offset, wlen, (m & kMetaOov) != 0u)) : 0.0);
    }
  }
  wave_sync();
  // (exact ties -- of the last-frame scores inside a group, of the groups' merged scores -- go to the larger key, not to the
  // earlier beam: the beams' order is not the same in the two kernels, see the select)
  double my_score = -1e300;
  unsigned long long my_key = 0;
  int my_first = 0x7fffffff;
#pragma unroll 1
  for (int i = lane; i < nb; i += 64) {
    const unsigned long long k = fkey[i];
    bool first = true;
    for (int j = 0; j < i; ++j) if (fkey[j] == k) { first = false; break; }
    if (!first) continue;
    double m = S.logit[cur][i];
    int rep = i;
    for (int j = i + 1; j < nb; ++j)
      if (fkey[j] == k) {
        m = fmax(m, S.logit[cur][j]);
        if (frank[j] < frank[rep] || (frank[j] == frank[rep] && S.key[cur][j] > S.key[cur][rep])) rep = j;
      }
    double ssum = 0;
    for (int j = i; j < nb; ++j) if (fkey[j] == k) ssum += exp(S.logit[cur][j] - m);
    const double merged = (fin[rep] - S.logit[cur][rep]) + m + log(ssum);
    if (merged > my_score || (merged == my_score && k > my_key)) { my_score = merged; my_key = k; my_first = i; }
  }
  const long long sbest = wave_max_i64(ord64(my_score));
  const long long kbest = wave_max_i64(ord64(my_score) == sbest ? (long long)(my_key ^ 0x8000000000000000ull) : (long long)0x8000000000000000ull);
  const unsigned long long wm = __ballot(ord64(my_score) == sbest && (long long)(my_key ^ 0x8000000000000000ull) == kbest);
  int bi_best = 0x7fffffff;
  for (unsigned long long q2 = wm; q2; q2 &= q2 - 1) bi_best = min(bi_best, __builtin_amdgcn_readlane(my_first, __ffsll((long long)q2) - 1));
  const double bs = unord64(sbest);

  // ---- trace back (as beam_wave.hip: rows through LDS kTbRows at a time, characters collected in LDS) ----
  unsigned int* rows = reinterpret_cast<unsigned int*>(S.tkey);
  unsigned short* chars = reinterpret_cast<unsigned short*>(&S.key[0][0]);
  int32_t* out = out_ids + (int64_t)b * frames_ld;
  const bool in_lds = frames <= kChars;
  int n = 0, cur_b = bi_best;
  bool lead = true;
  constexpr int kRowRegs = kTbRows * kMaxBeams / 4 / 64;
  uint4 rr[kRowRegs];
  const int nbatch = (frames + kTbRows - 1) / kTbRows;
  auto tb_request = [&](int j) __attribute__((always_inline)) {
    const int t_hi = frames - 1 - j * kTbRows, t_lo = max(0, t_hi - kTbRows + 1), nq = (t_hi - t_lo + 1) * (kMaxBeams / 4);
    const uint4* g = reinterpret_cast<const uint4*>(bp + (int64_t)t_lo * kMaxBeams);
#pragma unroll
    for (int k = 0; k < kRowRegs; ++k) rr[k] = 64 * k + lane < nq ? g[64 * k + lane] : make_uint4(0, 0, 0, 0);
  };
  auto tb_land = [&](int j) __attribute__((always_inline)) {
    uint4* dst = reinterpret_cast<uint4*>(rows + (j & 1) * kTbRows * kMaxBeams);
#pragma unroll
    for (int k = 0; k < kRowRegs; ++k) dst[64 * k + lane] = rr[k];
  };
  if (nbatch > 0) { tb_request(0); tb_land(0); }
  for (int j = 0; j < nbatch; ++j) {
    if (j + 1 < nbatch) tb_request(j + 1);
    wave_sync();
    const int t_hi = frames - 1 - j * kTbRows, t_lo = max(0, t_hi - kTbRows + 1);
    const unsigned int* rb = rows + (j & 1) * kTbRows * kMaxBeams;
    for (int tt = t_hi - t_lo; tt >= 0; --tt) {
      const unsigned int e = rb[tt * kMaxBeams + cur_b];
      const unsigned int ch = e & 255;
      if (ch) {
        const int id = (int)ch - 1;
        if (!(lead && id == space_id)) {
          lead = false;
          if (in_lds) chars[n] = (unsigned short)id;
          else if (lane == 0) out[frames_ld - 1 - n] = id;
          ++n;
        }
      }
      cur_b = (int)(e >> 8);
    }
    if (j + 1 < nbatch) tb_land(j + 1);
    wave_sync();
  }
  if (in_lds) {
    for (int j = lane; j < n; j += 64) out[j] = (int)chars[n - 1 - j];
  } else {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    const int off = frames_ld - n;
    if (off > 0) {
      for (int j0 = 0; j0 < n; j0 += 64) {
        const int j = j0 + lane;
        int v = 0;
        if (j < n) v = __hip_atomic_load(&out[off + j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if (j < n) out[j] = v;
      }
    }
  }
  if (lane == 0) {
    out_len[b] = S.overflow ? -1 : n;
    out_score[b] = (float)bs;
  }
}

template <int W>
int launch_group(const float* logp, int batch, int frames, int V1, int space_id, int beam_width, float token_min_logp,
                 float beam_prune_logp, const LmView& v, int use_lm, unsigned int* bp, unsigned long long* eoslog,
                 int32_t* out_ids, int32_t* out_len, float* out_score, hipStream_t st, const int32_t* row_frames) {
  static std::atomic<uint64_t> lds_opted{0};   // per device (dyn_lds_opt_in): the table alone is 72 KB
  const hipError_t attr = dyn_lds_opt_in(reinterpret_cast<const void*>(beam_group_kernel<W>), (int)sizeof(GroupLds<W>), lds_opted);
  if (attr != hipSuccess) return (int)attr;
  hipLaunchKernelGGL(beam_group_kernel<W>, dim3(batch), dim3(64 * W), sizeof(GroupLds<W>), st, logp, batch, frames, row_frames,
                     V1, space_id, beam_width, token_min_logp, beam_prune_logp, v, use_lm, bp, eoslog, out_ids, out_len,
                     out_score);
  return 0;
}

}  // namespace

// wavefronts an utterance of a batch gets: 4 up to kGroupMaxBatch = 64 utterances, 1 beyond (beam_wave.hip: four utterances per
// compute unit).  Rounds 4-5 drew the line at 16 ("a batch leaves the chip to the next acoustic pass").  Measured in round 6 at
// configs[3] (64 x 501 frames, beam 128 + LM, search of batch k under the acoustic pass of batch k + 1): the four-wavefront
// form holds 64 compute units for 1.31 ms where the one-wavefront form holds 16 for 2.13 ms -- 2.5 x the CU-time, and still the
// STEP is shorter (5.75 against 5.87 ms) and a job's last batch pays 0.70 ms instead of 1.43: every GEMM launch that meets a
// busy compute unit loses a partial round of tiles whatever the number of busy units, so what counts is how LONG the search
// is in the way, not how wide.  Beyond a quarter of the chip the one-wavefront form keeps the CU-time down.
constexpr int kGroupMaxBatch = 64;
int beam_group_width(int batch) {
  const int force = dev_switches().beam_group;   // devtools build: 0 | 1 = never the four-wavefront form, 4 = always
  if (force == 0 || force == 1) return 1;
  if (force == 4) return 4;
  return batch <= kGroupMaxBatch ? 4 : 1;
}

int launch_beam_search_group(const float* logp, int batch, int frames, int V1, int space_id, int beam_width,
                             float token_min_logp, float beam_prune_logp, const BeamLm* lm, unsigned int* bp,
                             int32_t* out_ids, int32_t* out_len, float* out_score, hipStream_t st,
                             const int32_t* row_frames) {
  const int W = beam_group_width(batch);
  if (W == 1) return launch_beam_search_wave(logp, batch, frames, V1, space_id, beam_width, token_min_logp, beam_prune_logp, lm,
                                             bp, out_ids, out_len, out_score, st, row_frames);
  unsigned long long* eoslog = reinterpret_cast<unsigned long long*>(bp + (size_t)batch * frames * kMaxBeams);
  const LmView v = make_lm_view(lm);
  const int use_lm = lm ? 1 : 0;
  return launch_group<4>(logp, batch, frames, V1, space_id, beam_width, token_min_logp, beam_prune_logp, v, use_lm, bp, eoslog,
                         out_ids, out_len, out_score, st, row_frames);
}

}  // namespace vasr
