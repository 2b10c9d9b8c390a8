// CTC prefix beam search with optional back-off n-gram LM, on the device (gfx950).
//
// Replaces BeamSearchDecoderWithLM.forward (reference nemo/collections/asr/beam_search_decoder.py:95-102), which
// hands exp(log_probs[0]) to pyctcdecode on the host, one utterance at a time.  pyctcdecode / kenlm are third-party
// and absent (parity unpinned); the algorithm restated here is oracle/beam_oracle.py (file header there).
//
// One workgroup (512 threads) per utterance walks the frames; per frame
//   1. candidate characters {c : logp >= token_min_logp} U {argmax}: one wavefront, ballots only.  The merge table
//      holds kMaxFill (beam, character) pairs; when a frame has more (flat posteriors: beams x candidates > 1434) the
//      candidates are taken in several PASSES of steps 2-3.  That is exact, not an approximation: the merge key carries
//      the candidate as "last character", so pairs of different candidates never merge, and the survivors of the
//      earlier passes are carried in registers into the next pass's selection (top-k of a union = top-k of the
//      partial top-ks; the prune threshold only rises from pass to pass and the last pass sees the global maximum);
//   2. every (beam, character) pair is hashed -- key = hash(prefix string incl. committed spaces, last character) --
//      into an LDS open-addressing table: identical prefixes MERGE by log-sum-exp (fp64 max via ordered-int
//      atomicMax, then a 2^-44 fixed-point atomicAdd of exp(score - max): associative, hence deterministic);
//   3. every thread pulls its own 4 table slots into registers; a word committed by ' ' is scored with the n-gram LM
//      (hashed tables in HBM, back-off walk), partial words get pyctcdecode's OOV penalty; beams below max-10 are
//      dropped; the top `beam_width` are kept by a radix select on the ordered bits of the combined score (8-bit
//      digits, parallel bucket search, stops as soon as a bucket is taken whole; no sort);
//   4. survivors publish (slot, merged logit) at their rank, the new beams are built one per thread and a
//      back-pointer row is written for the final trace-back.
// What the frame loop is bound by, and what the structure answers (measured with VASR_BEAM_PROF cycle counters):
// dependent LDS round trips and barriers, not arithmetic.  Hence DPP scans instead of ds_bpermute shuffles, LDS-only
// barriers (s_waitcnt lgkmcnt(0) + s_barrier: __syncthreads() also waits for the back-pointer stores and the
// log-prob prefetch), per-thread slots in registers between phases, and a loop body kept at 32 KB of code -- fully
// unrolled it was 68 KB, more than the instruction cache two CUs share.  B = 64 x 501 frames, beam 128: 10.7 -> 4.0 ms.
#include <cstdlib>
#include <type_traits>

#include "vasr_internal.h"
#include "beam_common.h"

namespace vasr {

namespace {
using namespace beam_detail;

#ifndef VASR_BEAM_THREADS
#define VASR_BEAM_THREADS 512
#endif
constexpr int kThreads = VASR_BEAM_THREADS;   // workgroup size: 256, 512 or 1024
constexpr int kWaves = kThreads / 64;
// Merge-table slots: a template parameter of the kernel (a power of two, a multiple of the workgroup size), chosen per
// launch from the beam width -- clearing and sweeping the table is a fixed cost per frame, so a narrow beam is faster with a
// small table and a wide one with few passes (launch_beam_search).  kMaxFill(slots) pairs per pass keep it <= 70 % full.
constexpr int max_fill(int slots) { return slots * 7 / 10; }   // 1433 at 2048 slots
struct Beam {
  unsigned long long key;    // hash of the prefix characters, committed spaces included
  unsigned long long whash;  // rolling hash of the current partial word (label ids)
  double logit;
  float lm_text;             // LM score of the committed words
  int last;                  // last emitted class (blank = V, none = -1)
  int wlen;                  // characters in the partial word
  int ctx[kMaxCtx];          // LM history, most recent last, -1 = empty
  int cached;                // "text + pending word" already sits in pyctcdecode's LM score cache (see eoslog below)
};

// Merge table, structure-of-arrays in LDS (consecutive threads touch consecutive words: no bank conflicts):
//   key   0 = empty                                  mx   ordered bits of the max score, later of the combined score
//   sum   fixed-point sum of exp(score - max), later the bits of the merged logit
//   src   (beam << 8) | class
// After the expand phase every thread keeps its kSlots / kThreads slots (i = tid + kThreads j) in registers for
// scoring and selection.
struct Slots {
  unsigned long long* key; long long* mx; unsigned long long* sum; int* src;
};
constexpr size_t slot_bytes(int slots) { return (size_t)slots * (8 + 8 + 8 + 4); }

// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains the vector-memory counter, i.e. it
// waits for the back-pointer stores of the previous frame to be acknowledged and for the prefetched log-prob row
// to arrive -- a full HBM round trip per frame that nothing in the workgroup depends on.
__device__ inline void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// Exclusive scan over the workgroup with ONE barrier: the per-wavefront totals go to one of two scratch rows that
// alternate from call to call (`flip`), so a row is rewritten only after another barrier has passed.
__device__ inline int block_scan_excl(int v, int* scratch, int* total, int& flip) {
  // kWaves wavefronts; scratch holds 2 x kWaves ints
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int x = wave_scan_incl(v);
  int* row = scratch + flip * kWaves;
  flip ^= 1;
  if (lane == 63) row[wave] = x;
  lds_barrier();
  int base = 0, all = 0;
#pragma unroll
  for (int w = 0; w < kWaves; ++w) { const int c = row[w]; all += c; if (w < wave) base += c; }
  *total = all;
  return base + x - v;
}

// grid (B), block kThreads
template <int kSlots>
__global__ __launch_bounds__(kThreads) void beam_search_kernel(const float* __restrict__ logp, int frames_ld,
                                                          const int32_t* __restrict__ row_frames, int V1,
                                                          int space_id, int beam_width, float token_min_logp,
                                                          float beam_prune_logp, LmView lm, int use_lm,
                                                          unsigned int* __restrict__ bp_all,
                                                          unsigned long long* __restrict__ eoslog_all,
                                                          int32_t* __restrict__ out_ids, int32_t* __restrict__ out_len,
                                                          float* __restrict__ out_score) {
  constexpr int kMaxFill = max_fill(kSlots);
  constexpr size_t kSlotBytes = slot_bytes(kSlots);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  Slots sl;
  sl.key = reinterpret_cast<unsigned long long*>(smem);
  sl.mx = reinterpret_cast<long long*>(sl.key + kSlots);
  sl.sum = reinterpret_cast<unsigned long long*>(sl.mx + kSlots);
  sl.src = reinterpret_cast<int*>(sl.sum + kSlots);
  Beam* beams = reinterpret_cast<Beam*>(smem + kSlotBytes);
  Beam* nbeams = beams + kMaxBeams;   // the two buffers swap roles every frame
  double* lp = reinterpret_cast<double*>(nbeams + kMaxBeams);  // [kMaxClasses]
  int* cand = reinterpret_cast<int*>(lp + kMaxClasses);        // [kMaxClasses]
  int* hist = cand + kMaxClasses;                              // [256]
  int* misc = hist + 256;                                      // [16] + [16] block-scan scratch
  long long* best = reinterpret_cast<long long*>(misc + 16 + 2 * kWaves + (kWaves & 1) * 0);   // [1] (+1 pad)
  // the survivor at each rank, as the record a new beam is built from (and the next pass's selection carries on)
  long long* sel_lgt = best + 2;                                // [kMaxBeams] bits of the merged logit
  long long* sel_tot = sel_lgt + kMaxBeams;                     // [kMaxBeams] ordered bits of the combined score
  float* sl_lmd = reinterpret_cast<float*>(sel_tot + kMaxBeams);   // [kSlots] LM score of the word a slot commits
  int* sl_wid = reinterpret_cast<int*>(sl_lmd + kSlots);            // [kSlots] its word id
  int* sel_src = sl_wid + kSlots;                                   // [kMaxBeams] (beam << 8) | class
  float* sel_lmd = reinterpret_cast<float*>(sel_src + kMaxBeams);   // [kMaxBeams]
  int* sel_wid = reinterpret_cast<int*>(sel_lmd + kMaxBeams);       // [kMaxBeams]
  unsigned short* pair_slot = reinterpret_cast<unsigned short*>(sel_wid + kMaxBeams);   // [kMaxFill + 2] slot of a pair

  const int tid = threadIdx.x, b = blockIdx.x, V = V1 - 1;
  // frames searched: all of them (the reference hands pyctcdecode every frame of its batch-1 tensor), or the row's own
  // count when the caller batches utterances of different lengths
  const int frames = row_frames ? max(0, min(frames_ld, row_frames[b])) : frames_ld;
  const float* lrow = logp + (int64_t)b * frames_ld * V1;
  unsigned int* bp = bp_all + (int64_t)b * frames_ld * kMaxBeams;
  // pyctcdecode caches the LM score of every text it has scored and consults that cache again when the pending word
  // is scored with </s> after the last frame: a text whose commit (text + ' ') was a candidate of ANY earlier frame
  // keeps its cached score WITHOUT </s>.  eoslog is that cache's key set: the hash of "text + pending word" of every
  // beam that met a frame with ' ' among the candidates (once per beam lineage, `cached`); <= frames x kMaxBeams entries.
  unsigned long long* eoslog = eoslog_all + (int64_t)b * frames_ld * kMaxBeams;

  if (tid == 0) {
    Beam s{};
    s.key = kFnvOffset; s.whash = kFnvOffset; s.logit = 0.0; s.lm_text = 0.f; s.last = -1; s.wlen = 0;
    for (int i = 0; i < kMaxCtx; ++i) s.ctx[i] = -1;
    if (use_lm) s.ctx[kMaxCtx - 1] = lm.bos;
    s.cached = 0;
    beams[0] = s;
    misc[0] = 1;  // live beams
    misc[9] = 0;  // not every live beam ends in blank
    misc[11] = 0; // entries in eoslog
  }
  lds_barrier();

#ifdef VASR_BEAM_PROF   // dev build: per-section cycle totals of workgroup 0 (tests/devtools/bench_beam.py with VASR_LIB_PATH)
  long long prof[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, pt = clock64();
#define BEAM_TICK(k) { const long long now = clock64(); prof[k] += now - pt; pt = now; }
#else
#define BEAM_TICK(k)
#endif
  constexpr int kSpt = kSlots / kThreads;       // table slots owned by a thread: i = tid + kThreads j
  // next frame's log-probs of classes tid and tid + 64 (wavefront 0 only), requested one frame ahead
  float lrow_next = tid < 64 && tid < V1 ? lrow[tid] : 0.f, lrow_next1 = tid < 64 && tid + 64 < V1 ? lrow[tid + 64] : 0.f;
  int flip = 0;   // block_scan_excl scratch row
  for (int t = 0; t < frames; ++t) {
    const int nb = misc[0];
    BEAM_TICK(5)
    // ---- 1. log-probs (pyctcdecode: log(clip(p, 1e-15, 1))) and candidate characters ----
    // log(clip(exp(x), 1e-15, 1)) = clip(x, log 1e-15, 0): the same value without two fp64 transcendentals
    // Wavefront 0 owns the log-probs (two classes per lane) and picks the candidates while the others clear the merge
    // table: one barrier for both.
    if (tid == 0) { best[0] = ord64(-1e300); best[1] = 0; }   // running max; OR of (key ^ best key) over the live keys
    const int all_blank = misc[9];   // every live beam ends in blank (set by the frame that built them)
    if (tid < 256) hist[tid] = 0;
#pragma unroll
    for (int j = 0; j < kSpt; ++j) { const int i = tid + kThreads * j; sl.key[i] = 0; sl.mx[i] = ord64(-1e300); sl.sum[i] = 0; }
    BEAM_TICK(8)
    BEAM_TICK(9)
    if (tid < 64) {
      // Candidates with ballots only (V1 <= 128) -- no LDS, no atomics,
      // fixed (class) order: wanted = {v >= token_min_logp} U {arg-max}; if more than `cap` are wanted, the cap largest
      // (ties: lower class first) are found by a bitwise threshold search on the order-preserving integer image of v.
      const int c0 = tid, c1 = tid + 64;
      const double d0 = fmin(fmax((double)lrow_next, -34.538776394910684), 0.0);
      const double d1 = fmin(fmax((double)lrow_next1, -34.538776394910684), 0.0);
      if (c0 < V1) lp[c0] = d0;
      if (c1 < V1) lp[c1] = d1;
      if (t + 1 < frames) {
        if (c0 < V1) lrow_next = lrow[(int64_t)(t + 1) * V1 + c0];
        if (c1 < V1) lrow_next1 = lrow[(int64_t)(t + 1) * V1 + c1];
      }
      const float v0 = c0 < V1 ? (float)d0 : 0.f, v1 = c1 < V1 ? (float)d1 : 0.f;
      auto okey = [](float v) { const unsigned b = __float_as_uint(v); return (b & 0x80000000u) ? ~b : (b | 0x80000000u); };
      const unsigned key0 = c0 < V1 ? okey(v0) : 0u, key1 = c1 < V1 ? okey(v1) : 0u;
      const unsigned kmax = wave_max_u32(max(key0, key1));
      const unsigned long long a0 = __ballot(c0 < V1 && key0 == kmax), a1 = __ballot(c1 < V1 && key1 == kmax);
      const int amax = a0 ? __ffsll((long long)a0) - 1 : 64 + __ffsll((long long)a1) - 1;
      bool k0 = c0 < V1 && (v0 >= token_min_logp || c0 == amax);
      bool k1 = c1 < V1 && (v1 >= token_min_logp || c1 == amax);
      const unsigned long long lt = (1ull << tid) - 1ull;
      const unsigned long long m0 = __ballot(k0), m1 = __ballot(k1);
      if (k0) cand[__popcll(m0 & lt)] = c0;
      if (k1) cand[__popcll(m0) + __popcll(m1 & lt)] = c1;
      if (tid == 0) {
        misc[1] = __popcll(m0) + __popcll(m1);
        misc[10] = space_id < 64 ? (int)(m0 >> space_id & 1) : (space_id < 128 ? (int)(m1 >> (space_id - 64) & 1) : 0);
      }
    }
    BEAM_TICK(10)
    lds_barrier();
    const int nc_all = misc[1];
    BEAM_TICK(0)
    // A frame whose only candidate is blank, met by beams that all end in blank already, changes nothing but the
    // scores, and those by the same amount: prefixes and last characters stay distinct (no merge), the LM parts and
    // every score difference stay what the previous frame's prune and selection saw.  A trained CTC model emits long
    // runs of such frames; the first of a run goes the general way (beams that differ only in their last character
    // merge there), the rest take this exit.  s.logit + lp is the sum the general path would have stored.
#ifdef VASR_BEAM_NO_BLANK_EXIT   // dev build for A/B runs
    if (false) {
#else
    if (nc_all == 1 && all_blank && cand[0] == V) {
#endif
      if (tid < nb) {
        beams[tid].logit += lp[V];
        bp[(int64_t)t * kMaxBeams + tid] = (unsigned)tid << 8;
      }
      lds_barrier();   // lp is rewritten by wavefront 0 at the top of the next frame
      continue;
    }
    if (tid == 0) misc[9] = 1;   // cleared below by any new beam that ends in a character
    const int has_space = misc[10];
    if (use_lm && has_space && tid < nb) {   // this frame puts "text + pending word" of every live beam into the LM cache
      const Beam& s = beams[tid];
      if (s.wlen > 0 && !s.cached) eoslog[atomicAdd(&misc[11], 1)] = hmix(s.key, (unsigned long long)space_id) | 1ull;
    }
    const int cap = max(1, kMaxFill / nb);   // candidates per pass: nb * cap pairs fit the merge table
    constexpr int kSel = kSpt + 1;           // a thread's table slots + the survivor it carries from the earlier passes
    long long c_tot = ord64(-1e300), c_lgt = 0;
    int c_src = 0, c_wid = 0, n_sel = 0;
    float c_lmd = 0.f;
#pragma unroll 1
    for (int c_lo = 0; c_lo < nc_all; c_lo += cap) {
    const int nc = min(cap, nc_all - c_lo);
    if (c_lo > 0) {
      if (tid == 0) best[1] = 0;
#pragma unroll
      for (int j = 0; j < kSpt; ++j) { const int i = tid + kThreads * j; sl.key[i] = 0; sl.mx[i] = ord64(-1e300); sl.sum[i] = 0; }
      lds_barrier();
    }
    // ---- 2. expand: every (beam, character) pair claims / finds its slot and raises the slot's max, then (after a
    //         barrier) adds exp(score - max).  Deliberately NOT unrolled: the kernel must stay inside the 64 KB
    //         instruction cache it shares with the neighbouring CU (68 KB unrolled: every section fetch-bound). ----
#pragma unroll 1
    for (int p = tid; p < nb * nc; p += kThreads) {
      const int bi = p / nc, c = cand[c_lo + p % nc];
      const Beam& s = beams[bi];
      unsigned long long key = s.key;
      if (!(c == V || c == s.last)) {
        if (c == space_id) { if (s.wlen > 0) key = hmix(key, (unsigned long long)c); }
        else key = hmix(key, (unsigned long long)c);
      }
      unsigned long long k = hmix(key, (unsigned long long)(c + 7)) | 1ull;   // (prefix, last char)
      // home slot and probe stride from the upper bits (bit 0 of k is forced to 1: the low bits would reach only the
      // odd slots); an odd stride visits every slot of the power-of-two table -- double hashing, no primary clustering
      int i = (int)((k >> 17) & (kSlots - 1));
      const int stride = (int)((k >> 40) & (kSlots - 1)) | 1;
      while (true) {
        const unsigned long long e = sl.key[i];
        if (e == k) break;
        if (e == 0) {
          const unsigned long long old = atomicCAS(&sl.key[i], 0ull, k);
          if (old == 0ull) { sl.src[i] = (bi << 8) | c; break; }
          if (old == k) break;
        }
        i = (i + stride) & (kSlots - 1);
      }
      atomicMax(&sl.mx[i], ord64(s.logit + lp[c]));
      pair_slot[p] = (unsigned short)i;
    }
    lds_barrier();
#pragma unroll 1
    for (int p = tid; p < nb * nc; p += kThreads) {
      const int i = pair_slot[p];
      const double score = beams[p / nc].logit + lp[cand[c_lo + p % nc]];
      // exp(score - max) through the hardware 2^x (v_exp_f32, 1 ulp): the library's fp64 exp was most of this phase
      // (150+ fp64 instructions per pair, three rounds of them per frame at 1434 pairs), and the sum only has to carry
      // the 1e-7 a float term gives -- the log of it is compared at 1e-3.  exp2(0) is exactly 1, so a slot with a
      // single contributor still holds exactly 2^44.
      const float e = __builtin_amdgcn_exp2f((float)((score - unord64(sl.mx[i])) * 1.4426950408889634));
      atomicAdd(&sl.sum[i], (unsigned long long)((double)e * kFix));
    }
    lds_barrier();
    BEAM_TICK(1)
    // ---- 3. the thread's own slots move into registers; LM scoring of committed words, combined score ----
    long long tot[kSel];      // ordered bits of the combined score
    long long lgt[kSel];      // bits of the merged logit
    unsigned live = 0;        // bit j: slot j is occupied (later: and survives the prune); bit kSpt: the carried survivor
    tot[kSpt] = tid < n_sel ? c_tot : ord64(-1e300);
    lgt[kSpt] = c_lgt;
    if (tid < n_sel) live |= 1u << kSpt;
    {
      unsigned long long k8[kSpt], s8[kSpt];
      long long m8[kSpt];
      int src[kSpt];
#pragma unroll
      for (int j = 0; j < kSpt; ++j) {
        const int i = tid + kThreads * j;
        k8[j] = sl.key[i]; m8[j] = sl.mx[i]; s8[j] = sl.sum[i]; src[j] = sl.src[i];
      }
      unsigned need = 0;      // slots whose candidate commits a word: LM score wanted
      float lmt[kSpt];        // LM part of the combined score
#pragma unroll
      for (int j = 0; j < kSpt; ++j) {
        lmt[j] = 0.f;
        if (k8[j] == 0) continue;
        live |= 1u << j;
        if (use_lm) {
          const int bi = src[j] >> 8, c = src[j] & 255;
          const Beam& s = beams[bi];
          const bool stay = (c == V || c == s.last);
          if (!stay && c == space_id && s.wlen > 0) need |= 1u << j;
          const int wlen_new = stay ? s.wlen : (c == space_id ? 0 : s.wlen + 1);
          lmt[j] = s.lm_text + partial_penalty(lm.unk_offset, wlen_new);
        }
      }
      // one copy of the n-gram walk for all slots of the thread
      for (unsigned nm = need; nm; nm &= nm - 1) {
        const int i = tid + kThreads * (__ffs(nm) - 1);
        const Beam& s = beams[sl.src[i] >> 8];
        int w;
        sl_lmd[i] = lm_word_score(lm, s.ctx, s.whash, false, &w);
        sl_wid[i] = w;
      }
      long long my_best = ord64(-1e300);
#pragma unroll
      for (int j = 0; j < kSpt; ++j) {
        tot[j] = ord64(-1e300); lgt[j] = 0;
        if (!(live >> j & 1)) continue;
        if (need >> j & 1) lmt[j] += sl_lmd[tid + kThreads * j];   // written by this thread just above
        // a slot with a single contributor holds exactly exp(0) * 2^44: no logarithm needed
        const double logit = unord64(m8[j]) + (s8[j] == (unsigned long long)kFix ? 0.0 : log_ge1((double)s8[j] * (1.0 / kFix)));
        tot[j] = ord64(logit + (double)lmt[j]);
        lgt[j] = __double_as_longlong(logit);
        my_best = max(my_best, tot[j]);
      }
      // one LDS atomic per wavefront instead of one per candidate (they all hit the same address)
      my_best = wave_max_i64(my_best);
      if ((tid & 63) == 0) atomicMax(best, my_best);
    }
    lds_barrier();
    BEAM_TICK(2)
    // ---- 4. prune (max + beam_prune_logp) and radix-select the top beam_width by combined score ----
    const long long thr_prune = ord64(unord64(*best) + (double)beam_prune_logp);
    unsigned long long u8[kSel];   // keys as unsigned radix digits
#pragma unroll
    for (int j = 0; j < kSel; ++j) {
      if (tot[j] < thr_prune) live &= ~(1u << j);
      u8[j] = (unsigned long long)tot[j] ^ 0x8000000000000000ull;
    }
    unsigned long long prefix = 0, mask = 0;
    int want = beam_width;   // how many still to take among keys matching the prefix
    {
      // Scores of live beams lie within beam_prune_logp of the best, so their keys share the sign, the exponent and
      // usually the top mantissa bits: the leading digits all keys have in common are skipped instead of costing a
      // pass each.  The OR of (key ^ best key) tells where they first differ; it rides on the barrier of the scan.
      const unsigned long long ubest = (unsigned long long)*best ^ 0x8000000000000000ull;
      unsigned long long d = 0;
#pragma unroll
      for (int j = 0; j < kSel; ++j) d |= (live >> j & 1) ? (u8[j] ^ ubest) : 0ull;
      d = ((unsigned long long)wave_or_u32((unsigned)(d >> 32)) << 32) | wave_or_u32((unsigned)d);
      if ((tid & 63) == 0 && d) atomicOr(reinterpret_cast<unsigned long long*>(best + 1), d);
      int tot_live;
      block_scan_excl(__popc(live), misc + 16, &tot_live, flip);
      if (tot_live > beam_width) {
        const unsigned long long diff = (unsigned long long)best[1];
        const int same = diff ? __clzll((long long)diff) / 8 : 8;       // leading bytes common to every live key
        if (same > 0) { mask = ~0ull << (64 - 8 * same); if (same == 8) mask = ~0ull; prefix = ubest & mask; }
#pragma unroll 1
        // three barriers per digit: the histogram is all zero on entry (cleared with the table at the top of the frame)
        // and every thread re-zeroes its own bin right after reading it
        for (int shift = 56 - 8 * same; shift >= 0; shift -= 8) {
#pragma unroll
          for (int j = 0; j < kSel; ++j)
            if ((live >> j & 1) && (u8[j] & mask) == prefix) atomicAdd(&hist[(int)((u8[j] >> shift) & 255)], 1);
          lds_barrier();
          // the bucket holding the want-th largest key, searched from the top by all threads at once: thread tid owns
          // bin 255 - tid and a block scan gives it the number of keys in the bins above
          int mine = 0;
          if (tid < 256) { mine = hist[255 - tid]; hist[255 - tid] = 0; }
          int all;
          const int above = block_scan_excl(mine, misc + 16, &all, flip);
          if (above < want && want <= above + mine) {
            misc[2] = 255 - tid; misc[3] = want - above;
            misc[8] = (mine == want - above);   // the whole bucket is taken: no need to refine further
          }
          lds_barrier();
          prefix |= (unsigned long long)misc[2] << shift;
          mask |= 0xFFull << shift;
          want = misc[3];
          if (misc[8]) break;
        }
      }
    }
    BEAM_TICK(3)
    // selected: key > threshold prefix, plus the first `want` (in thread-major slot order) equal to it
    unsigned gt = 0, eq = 0;
#pragma unroll
    for (int j = 0; j < kSel; ++j) {
      if (!(live >> j & 1)) continue;
      const unsigned long long u = u8[j] & mask;
      if (mask == 0 || u > prefix) gt |= 1u << j; else if (u == prefix) eq |= 1u << j;
    }
    int tot_pk;   // both counts through one scan: < 2048 each, packed 16 + 16 bits
    const int off_pk = block_scan_excl(__popc(gt) | (__popc(eq) << 16), misc + 16, &tot_pk, flip);
    const int off_gt = off_pk & 0xffff, off_eq = off_pk >> 16, tot_gt = tot_pk & 0xffff, tot_eq = tot_pk >> 16;
    const int take_eq = mask == 0 ? 0 : min(want, tot_eq);
    const int n_new = min(kMaxBeams, tot_gt + take_eq);
    BEAM_TICK(6)
    {
      // survivors publish their record at their rank (nobody reads the records while they are written: the carried
      // ones live in registers); after the last pass the new beams are built from them, one per thread
      int ig = off_gt, ie = off_eq;
#pragma unroll
      for (int j = 0; j < kSel; ++j) {
        int dst = -1;
        if (gt >> j & 1) dst = ig++;
        else if (eq >> j & 1) { if (ie < take_eq) dst = tot_gt + ie; ++ie; }
        if (dst >= 0 && dst < kMaxBeams) {
          if (j < kSpt) {
            const int i = tid + kThreads * j;
            sel_src[dst] = sl.src[i]; sel_lmd[dst] = sl_lmd[i]; sel_wid[dst] = sl_wid[i];
          } else {
            sel_src[dst] = c_src; sel_lmd[dst] = c_lmd; sel_wid[dst] = c_wid;
          }
          sel_lgt[dst] = lgt[j]; sel_tot[dst] = tot[j];
        }
      }
    }
    n_sel = n_new;
    lds_barrier();
    if (c_lo + cap < nc_all && tid < n_sel) {   // more candidates to come: this thread carries the survivor of its rank
      c_tot = sel_tot[tid]; c_lgt = sel_lgt[tid]; c_src = sel_src[tid]; c_lmd = sel_lmd[tid]; c_wid = sel_wid[tid];
    }
    }   // passes
    const int n_new = n_sel;
    if (tid < n_new) {
      const int bi = sel_src[tid] >> 8, c = sel_src[tid] & 255;
      const Beam& s = beams[bi];
      Beam n = s;
      const bool stay = (c == V || c == s.last);
      unsigned int appended = 0;
      if (!stay) {
        if (c == space_id) {
          if (s.wlen > 0) {
            n.key = hmix(s.key, (unsigned long long)c);
            appended = c + 1;
            if (use_lm) {
              n.lm_text = s.lm_text + sel_lmd[tid];
              for (int q = 0; q < kMaxCtx - 1; ++q) n.ctx[q] = s.ctx[q + 1];
              n.ctx[kMaxCtx - 1] = sel_wid[tid];
            }
            n.wlen = 0; n.whash = kFnvOffset;
          }
        } else {
          n.key = hmix(s.key, (unsigned long long)c);
          n.whash = hmix(s.whash, (unsigned long long)c);
          n.wlen = s.wlen + 1;
          appended = c + 1;
        }
      }
      n.last = c;
      // same text and pending word as the parent: in the LM cache if the parent was, or if this frame put it there
      n.cached = stay ? (s.cached | (has_space && s.wlen > 0)) : 0;
      if (c != V) misc[9] = 0;
      n.logit = __longlong_as_double(sel_lgt[tid]);
      nbeams[tid] = n;
      bp[(int64_t)t * kMaxBeams + tid] = ((unsigned)bi << 8) | appended;
    }
    BEAM_TICK(7)
    if (tid == 0) misc[0] = n_new;
    { Beam* x = beams; beams = nbeams; nbeams = x; }
    lds_barrier();
    BEAM_TICK(4)
  }
#ifdef VASR_BEAM_PROF
  if (tid == 0 && b == 0)
    printf("beam prof (cycles/frame): candidates %lld expand %lld lm %lld select %lld count+scan %lld compact %lld tail %lld other %lld  clear %lld bar %lld rank %lld live beams %d\n",
           prof[0] / frames, prof[1] / frames, prof[2] / frames, prof[3] / frames, prof[6] / frames, prof[7] / frames,
           prof[4] / frames, prof[5] / frames, prof[8] / frames, prof[9] / frames, prof[10] / frames, misc[0]);
#endif

  // ---- final: commit pending words (LM score with </s>), merge identical texts, pick the best ----
  const int nb = misc[0];
  // Is "text + pending word" in pyctcdecode's LM cache (then its cached score, WITHOUT </s>, is what the final pass uses)?
  // Known for beams whose own lineage put it there (`cached`); the others look their hash up in eoslog: their hashes go
  // into the (now idle) merge table, every thread walks a stride of the log and marks the hashes it meets.
  int in_cache = 0;
  if (use_lm) {
    const int nlog = misc[11];
    __syncthreads();   // the log's stores have left the CU
#pragma unroll
    for (int j = 0; j < kSpt; ++j) { const int i = tid + kThreads * j; sl.key[i] = 0; sl.src[i] = 0; }
    lds_barrier();
    int myslot = -1;
    if (tid < nb && beams[tid].wlen > 0) {
      in_cache = beams[tid].cached;
      if (!in_cache && nlog > 0) {
        const unsigned long long k = hmix(beams[tid].key, (unsigned long long)space_id) | 1ull;
        int i = (int)((k >> 17) & (kSlots - 1));
        while (true) {
          const unsigned long long old = atomicCAS(&sl.key[i], 0ull, k);
          if (old == 0ull || old == k) break;
          i = (i + 1) & (kSlots - 1);
        }
        myslot = i;
      }
    }
    lds_barrier();
    for (int q = tid; q < nlog; q += kThreads) {
      const unsigned long long k = __hip_atomic_load(&eoslog[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      for (int i = (int)((k >> 17) & (kSlots - 1));; i = (i + 1) & (kSlots - 1)) {
        const unsigned long long e = sl.key[i];
        if (e == k) { sl.src[i] = 1; break; }
        if (e == 0) break;
      }
    }
    lds_barrier();
    if (myslot >= 0) in_cache = sl.src[myslot];
    lds_barrier();   // sl.key / sl.mx are reused below
  }
  double* fin = lp;  // [kMaxBeams] combined score per beam
  unsigned long long* fkey = sl.key;  // [kMaxBeams]
  double* frank = reinterpret_cast<double*>(sl.mx);   // [kMaxBeams]
  if (tid < nb) {
    const Beam& s = beams[tid];
    double total = s.logit;
    if (use_lm) {
      float lmv = s.lm_text;
      int wid;
      if (s.wlen > 0) lmv += lm_word_score(lm, s.ctx, s.whash, !in_cache, &wid);
      total += (double)lmv;
    }
    fin[tid] = total;
    fkey[tid] = s.wlen > 0 ? hmix(s.key, (unsigned long long)space_id) : s.key;
    // the beam's combined score of the last frame: pyctcdecode keeps its beams sorted by it, and that order decides
    // below which member of a merged group provides the LM part
    frank[tid] = s.logit + (use_lm ? (double)(s.lm_text + partial_penalty(lm.unk_offset, s.wlen)) : 0.0);
  }
  __syncthreads();
  if (tid == 0) {
    // Merge by text: log-sum-exp of the LOGIT scores, as pyctcdecode does.  "abc" with the word still pending and
    // "abc " with it committed are the same final text but not the same LM part (only the pending word is scored with
    // </s>): pyctcdecode's _merge_beams overwrites the group's entry with every further member it meets while walking
    // its score-sorted beam list, so the member with the LOWEST last-frame score provides the LM part.
    int bi = 0;
    double bs = -1e300;
    for (int i = 0; i < nb; ++i) {
      bool first = true;
      for (int j = 0; j < i; ++j) if (fkey[j] == fkey[i]) { first = false; break; }
      if (!first) continue;
      double m = beams[i].logit;
      int rep = i;
      for (int j = i + 1; j < nb; ++j)
        if (fkey[j] == fkey[i]) { m = fmax(m, beams[j].logit); if (frank[j] < frank[rep]) rep = j; }
      double ssum = 0;
      for (int j = i; j < nb; ++j) if (fkey[j] == fkey[i]) ssum += exp(beams[j].logit - m);
      const double merged = (fin[rep] - beams[rep].logit) + m + log(ssum);
      if (merged > bs) { bs = merged; bi = i; }
    }
    // trace back
    int n = 0, cur = bi;
    int32_t* out = out_ids + (int64_t)b * frames_ld;
    for (int t = frames - 1; t >= 0; --t) {
      const unsigned int e = bp[(int64_t)t * kMaxBeams + cur];
      const unsigned int ch = e & 255;
      if (ch) out[n++] = (int)ch - 1;
      cur = (int)(e >> 8);
    }
    for (int i = 0; i < n / 2; ++i) { const int32_t x = out[i]; out[i] = out[n - 1 - i]; out[n - 1 - i] = x; }
    while (n > 0 && out[n - 1] == space_id) --n;  // normalise trailing whitespace
    out_len[b] = n;
    out_score[b] = (float)bs;
  }
}

}  // namespace

size_t beam_lds_bytes(int slots) {
  return slot_bytes(slots) + sizeof(Beam) * 2 * kMaxBeams + sizeof(double) * kMaxClasses + sizeof(int) * (kMaxClasses + 256 + 16 + 2 * kWaves) + 16 +
         (8 + 8 + 4 + 4 + 4) * kMaxBeams + 8 * slots + 2 * (max_fill(slots) + 2) + 16;
}

// Measured (MI355X, 64 x 501 frames, tools/gpu.sh beamslots): CTC-like posteriors, beam 20 / 50 / 100 / 128:
// 2.22 / 2.56 / 3.55 / 4.34 ms at 1024 slots, 2.63 / 2.99 / 3.83 / 4.50 at 2048; peaked ones 2.90 / 3.45 / 5.63 / 7.20 against
// 3.43 / 3.91 / 5.20 / 6.28.  (512 slots: another 5-7 % at beam <= 20, but 1.5x slower there on flat posteriors.)
int beam_slots_for(int beam_width) {
  static const int forced = dev_env("VASR_BEAM_SLOTS") ? atoi(dev_env("VASR_BEAM_SLOTS")) : 0;
  if (forced == 512 || forced == 1024 || forced == 2048) return forced;
  return beam_width > 64 ? 2048 : 1024;
}

int launch_beam_search(const float* logp, int batch, int frames, int V1, int space_id, int beam_width,
                        float token_min_logp, float beam_prune_logp, const BeamLm* lm, unsigned int* bp,
                        int32_t* out_ids, int32_t* out_len, float* out_score, hipStream_t st,
                        const int32_t* row_frames) {
  unsigned long long* eoslog = reinterpret_cast<unsigned long long*>(bp + (size_t)batch * frames * kMaxBeams);
  LmView v{};
  int use_lm = 0;
  if (lm) {
    use_lm = 1;
    v.vocab = static_cast<const uint4*>(lm->vocab); v.vcap = lm->vcap; v.ngram = static_cast<const uint4*>(lm->ngram);
    v.ncap = lm->ncap; v.order = lm->order; v.bos = lm->bos;
    v.vlg = 31 - __builtin_clz((unsigned)lm->vcap); v.nlg = 31 - __builtin_clz((unsigned)lm->ncap);
    v.eos = lm->eos; v.unk = lm->unk; v.alpha = lm->alpha; v.beta = lm->beta; v.unk_offset = lm->unk_offset;
  }
  auto go = [&](auto slots_tag) -> int {
    constexpr int slots = decltype(slots_tag)::value;
    auto kern = beam_search_kernel<slots>;
    const size_t lds = beam_lds_bytes(slots);
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (attr != hipSuccess) return (int)attr;
    hipLaunchKernelGGL(kern, dim3(batch), dim3(kThreads), lds, st, logp, frames, row_frames, V1, space_id, beam_width,
                       token_min_logp, beam_prune_logp, v, use_lm, bp, eoslog, out_ids, out_len, out_score);
    return 0;
  };
  const int slots = beam_slots_for(beam_width);
#ifdef VASR_DEVTOOLS
  if (slots == 512 && kThreads <= 512) return go(std::integral_constant<int, 512>{});
#endif
  if (slots == 1024) return go(std::integral_constant<int, 1024>{});
  return go(std::integral_constant<int, 2048>{});
}


}  // namespace vasr
