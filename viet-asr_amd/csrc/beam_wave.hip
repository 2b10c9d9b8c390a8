// CTC prefix beam search with optional back-off n-gram LM: ONE WAVEFRONT PER UTTERANCE (gfx950).
//
// Replaces BeamSearchDecoderWithLM.forward (reference nemo/collections/asr/beam_search_decoder.py:95-102), which hands
// exp(log_probs[0]) to pyctcdecode on the host, one utterance at a time -- batch 1, beam width 50 (app.py:27) or 100
// (infer.py:191) is the shape the reference SERVES.  pyctcdecode / kenlm are third-party and absent (parity unpinned);
// the algorithm restated here is oracle/beam_oracle.py (file header there).
//
// Rounds 1-3 ran one 512-thread workgroup per utterance (beam.hip, retired in round 5): every frame crossed
// ~20 workgroup barriers and every one of its eight wavefronts executed the whole ~5 000-instruction frame program --
// 13 us per general frame although a frame of a pruned search has 30-150 (beam, character) pairs, less than one per
// thread: the time was instruction issue and barrier latency, replicated eight times, not work.  This kernel gives an
// utterance to ONE wavefront: no s_barrier anywhere (LDS operations of one wavefront execute in order), scans,
// reductions and ranks are ballots / DPP, every per-frame cost is proportional to the pairs and merged prefixes the
// frame really has (claimed table slots go on a list: nothing sweeps or clears the whole table), the log-probs of the
// next frames are prefetched four deep, the LM score of a pending word is computed once per (text, word) lineage and
// inherited, and the final trace-back walks the back-pointer rows through LDS in batches instead of one dependent HBM
// round trip per frame.  An utterance needs 37.6 KB (38 528 bytes) of LDS (sizeof(WaveLds), pinned below) and 1 of the 16 wavefront slots a compute unit has, so four
// utterances share a CU (one workgroup of `upw` independent wavefronts): a batch of 64 occupies 16 compute units instead
// of 64 while the acoustic pass of the next batch runs on the rest.
//
// Per frame:
//   1. candidate characters {c : logp >= token_min_logp} U {argmax}: ballots (lane = class, two classes per lane);
//      a frame whose only candidate is blank, met by beams that all end in blank, shifts every score by the same
//      amount and changes nothing else: add, write an identity back-pointer row, next frame;
//   2. every (beam, character) pair is hashed -- key = hash(prefix string incl. committed spaces, last character) --
//      into an LDS open-addressing table where identical prefixes MERGE by log-sum-exp (fp64 max via ordered-int
//      atomicMax, then a 2^-44 fixed-point atomicAdd of exp(score - max): associative, hence deterministic).  A frame
//      with more pairs than the table takes (flat posteriors) runs its candidates in several PASSES -- pairs of
//      different candidates never merge (the key carries the candidate), the survivors of the earlier passes are
//      carried in registers into the next pass's selection (top-k of a union = top-k of the partial top-ks);
//   3. every merged prefix gets its combined score: a word committed by ' ' is scored with the n-gram LM (hashed
//      tables in HBM, back-off walk), partial words get pyctcdecode's OOV penalty; prefixes below max-10 are dropped;
//      the top `beam_width` are kept by a radix select on the ordered bits of the combined score (8-bit digits; skipped
//      when everything that survived the prune fits, which is the usual case);
//   4. survivors become the new beams (parent fields gathered from LDS), one back-pointer row per frame.
#include <cstdlib>
#include <type_traits>

#include "beam_common.h"

namespace vasr {

namespace {
using namespace beam_detail;

constexpr int kTab = 512;                 // merge-table slots per utterance
constexpr int kFill = kTab * 7 / 10;      // pairs per pass (<= 70 % load)
constexpr int kTbRows = 12;               // back-pointer rows per trace-back batch (6 KB, two batches in LDS)
constexpr int kLpFrames = 8;              // frames of log-probs per staging batch
constexpr int kLpRegs = kLpFrames * kMaxClasses / 64;   // floats a lane holds of the batch in flight
constexpr int kChars = 3072;              // characters of a transcript assembled in LDS (longer ones go through HBM)

// One utterance's working set in LDS.
struct WaveLds {
  // beams, two buffers that swap roles every frame (structure of arrays: lanes read consecutive words)
  unsigned long long key[2][kMaxBeams];    // hash of the prefix characters, committed spaces included
  unsigned long long whash[2][kMaxBeams];  // rolling hash of the pending word (label ids)
  double logit[2][kMaxBeams];
  float lm_text[2][kMaxBeams];             // LM score of the committed words
  // (last + 1) [7:0] (0 = none, blank = V + 1) | wlen [23:8] | cached [24] | commit_valid [25] | pending word is "OOV" [26]
  // (is_oov of pyctcdecode's score_partial_token: always with no unigram list, else "not a node of the character trie")
  unsigned int meta[2][kMaxBeams];
  int ctx[2][kMaxBeams][kMaxCtx];          // LM history, most recent last, -1 = empty
  float commit_lmd[2][kMaxBeams];          // LM score the pending word gets when ' ' commits it (valid: meta bit 25)
  int commit_wid[2][kMaxBeams];            // its word id
  // merge table:  tkey 0 = empty;  tmx ordered bits of the max score, later of the combined score;
  //               tsum fixed-point sum of exp(score - max), later the bits of the merged logit;  tsrc (beam << 8) | class
  unsigned long long tkey[kTab];
  long long tmx[kTab];
  unsigned long long tsum[kTab];
  int tsrc[kTab];
  // survivors of a pass that is not a frame's last, by rank (they come back as carried survivors); after the last frame:
  // the final text keys / last-frame scores of the beams
  long long sel_lgt[kMaxBeams], sel_tot[kMaxBeams];
  int sel_src[kMaxBeams];
  double fin[kMaxBeams];                   // combined final score per beam (final pass)
  unsigned long long cmix[kMaxClasses];    // per-class constant folded into a pair's key (the "last character" part)
  float lpq[kLpFrames * kMaxClasses];      // log-probs of the current batch of frames, [frame][class] as in memory
  unsigned char cand[kMaxClasses];
  int hist[256];
};
static_assert(sizeof(WaveLds) * 4 <= 160 * 1024, "four utterances per compute unit");
static_assert(sizeof(WaveLds) == 38528, "a new field changes the LDS per utterance: check that four still fit a compute unit, then update this number");
static_assert(2 * kTbRows * kMaxBeams * 4 <= (int)(sizeof(unsigned long long) * kTab * 3), "trace-back batches alias tkey + tmx + tsum");
static_assert(kChars * 2 <= (int)(sizeof(unsigned long long) * 2 * kMaxBeams * 3), "transcript characters alias the beam keys / hashes / logits");

constexpr unsigned kMetaCached = 1u << 24, kMetaCommit = 1u << 25, kMetaOov = 1u << 26;
constexpr int kSrcOov = 1 << 16;          // a pair record (beam << 8 | class) carries its child's "OOV" bit here
__device__ inline int meta_last(unsigned m) { return (int)(m & 0xffu) - 1; }
__device__ inline int meta_wlen(unsigned m) { return (int)((m >> 8) & 0xffffu); }
__device__ inline unsigned make_meta(int last, int wlen, unsigned flags) {
  return (unsigned)(last + 1) | ((unsigned)min(wlen, 0xffff) << 8) | flags;
}

// Orders the LDS traffic of the wavefront's phases for the COMPILER (the hardware executes one wavefront's LDS
// operations in order): lanes read what other lanes of the same wavefront wrote, which per-thread alias analysis
// cannot see.
__device__ inline void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ inline int lane_id() { return (int)(threadIdx.x & 63); }
__device__ inline int rank_in(unsigned long long mask) {   // set bits of `mask` below this lane
  return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
}
__device__ inline int wave_sum(int v) { return __builtin_amdgcn_readlane(wave_scan_incl(v), 63); }

// grid (ceil(B / upw)), block 64 * upw: wavefront w of workgroup g searches utterance g * upw + w, alone
__global__ __launch_bounds__(256) void beam_wave_kernel(const float* __restrict__ logp, int batch, int frames_ld,
                                                        const int32_t* __restrict__ row_frames, int V1, int space_id,
                                                        int beam_width, float token_min_logp, float beam_prune_logp,
                                                        LmView lm, int use_lm, unsigned int* __restrict__ bp_all,
                                                        unsigned long long* __restrict__ eoslog_all,
                                                        int32_t* __restrict__ out_ids, int32_t* __restrict__ out_len,
                                                        float* __restrict__ out_score) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = lane_id(), wv = (int)(threadIdx.x >> 6);
  const int b = (int)blockIdx.x * (int)(blockDim.x >> 6) + wv;
  if (b >= batch) return;                                   // (no barrier anywhere below: a wavefront may leave)
  WaveLds& S = *reinterpret_cast<WaveLds*>(smem + (size_t)wv * sizeof(WaveLds));
  const int V = V1 - 1;
  const bool trie = use_lm && lm.trie != nullptr;             // pyctcdecode's unigram set + character trie (".arpa" semantics)
  const int frames = row_frames ? max(0, min(frames_ld, row_frames[b])) : frames_ld;
  const float* lrow = logp + (int64_t)b * frames_ld * V1;
  unsigned int* bp = bp_all + (int64_t)b * frames_ld * kMaxBeams;
  // pyctcdecode's LM score cache, as its key set (see step 2b): hash of "text + pending word" of every beam that met a
  // frame with ' ' among the candidates, once per lineage
  unsigned long long* eoslog = eoslog_all + (int64_t)b * frames_ld * kMaxBeams;


  // ---- empty table, one beam: the empty prefix ----
  for (int i = lane; i < kTab; i += 64) { S.tkey[i] = 0; S.tmx[i] = ord64(-1e300); S.tsum[i] = 0; }
  for (int c = lane; c < kMaxClasses; c += 64) S.cmix[c] = hmix(hmix(kFnvOffset, (unsigned long long)(c + 7)), 0x9e3779b9ull);
  if (lane == 0) {
    S.key[0][0] = kFnvOffset; S.whash[0][0] = kFnvOffset; S.logit[0][0] = 0.0; S.lm_text[0][0] = 0.f;
    S.meta[0][0] = make_meta(-1, 0, 0);
    for (int i = 0; i < kMaxCtx; ++i) S.ctx[0][0][i] = -1;
    if (use_lm) S.ctx[0][0][kMaxCtx - 1] = lm.bos;
    S.commit_lmd[0][0] = 0.f; S.commit_wid[0][0] = 0;
  }
  wave_sync();
  int cur = 0, nb = 1, n_log = 0;
  bool all_blank = false;      // every live beam ends in blank (uniform)

  // Log-probs reach the frames through LDS, kLpFrames frames per batch (a contiguous run of kLpFrames * V1 floats), and
  // the batch after the current one is already on its way in registers.  Two things measured on the way: (1) a per-frame
  // register prefetch does not work -- s_waitcnt vmcnt counts in order, so waiting for the frame requested four frames
  // ago also waits for the request just issued and for the back-pointer stores: one HBM round trip per frame; (2) LLVM
  // SINKS plain loads to their first use, i.e. to the next batch boundary, and with a predicate per load they came out as
  // sixteen dependent round trips there (8 900 cycles per batch; volatile loads are worse: the back end drains the counter
  // after each).  Hence: unconditional loads of a clamped index, all in flight together.
  float q[kLpRegs];
  const int lp_batch = kLpFrames * V1;                        // floats per batch
  auto lp_request = [&](int t0) __attribute__((always_inline)) {
    const int n = min(lp_batch, (frames - t0) * V1);
    if (n <= 0) return;
    const float* src = lrow + (int64_t)t0 * V1;
#pragma unroll
    for (int k = 0; k < kLpRegs; ++k) q[k] = src[min(64 * k + lane, n - 1)];
  };
  lp_request(0);

  // One new beam at rank r from pair (parent bi, character c) with merged logit bits lgt: the parent's fields are gathered
  // from the current buffer, the child goes to the other one, one back-pointer word per rank and frame.
  auto build_child = [&](int t, int r, int src, long long lgt, bool has_space, bool& any_char) __attribute__((always_inline)) {
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
      // same text and pending word as the parent: in the LM cache if the parent was, or if this frame put it there; the
      // commit score of the pending word is inherited with them
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
    if (c != V) any_char = true;
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
    // ---- 1. candidate characters.  pyctcdecode works on log(clip(p, 1e-15, 1)) = clip(x, log 1e-15, 0); for every class
    //         that can be a candidate (x >= token_min_logp, or the arg-max) that is min(x, 0), a float: no fp64 here ----
    const int c0 = lane, c1 = lane + 64;
    if ((t & (kLpFrames - 1)) == 0) {                        // batch boundary: land the batch in LDS, request the next one
#pragma unroll
      for (int k = 0; k < kLpRegs; ++k) if (64 * k + lane < lp_batch) S.lpq[64 * k + lane] = q[k];
      lp_request(t + kLpFrames);
      wave_sync();
    }
    const float* lq = S.lpq + (t & (kLpFrames - 1)) * V1;
    const float v0 = c0 < V1 ? fminf(fmaxf(lq[c0], -34.538776f), 0.f) : 0.f, v1 = c1 < V1 ? fminf(fmaxf(lq[c1], -34.538776f), 0.f) : 0.f;
    auto okey = [](float v) { const unsigned q = __float_as_uint(v); return (q & 0x80000000u) ? ~q : (q | 0x80000000u); };
    // candidates = {x >= token_min_logp} U {arg-max}: when any class passes the threshold the arg-max is among them already
    // (round 5: the reduction for the arg-max is ~25 instructions of every frame's serial part)
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
    if (k0) S.cand[rank_in(m0)] = (unsigned char)c0;
    if (k1) S.cand[__popcll(m0) + rank_in(m1)] = (unsigned char)c1;
    const int nc_all = __popcll(m0) + __popcll(m1);
    const bool has_space = space_id < 64 ? (m0 >> space_id & 1) : (space_id < 128 ? (m1 >> (space_id - 64) & 1) : false);
    const bool only_blank = nc_all == 1 && (V < 64 ? (m0 >> V & 1) : (m1 >> (V - 64) & 1));
    wave_sync();

    // A frame whose only candidate is blank, met by beams that all end in blank already, changes nothing but the
    // scores, and those by the same amount: prefixes and last characters stay distinct (no merge), the LM parts and
    // every score difference stay what the previous frame's prune and selection saw.  A trained CTC model emits long
    // runs of such frames; the first of a run goes the general way (beams that differ only in their last character
    // merge there).  logit + lp is the sum the general path would have stored.
    if (only_blank && all_blank) {
      const double add = (double)fminf(lq[V], 0.f);
      for (int i = lane; i < nb; i += 64) {
        S.logit[cur][i] += add;
        bp[(int64_t)t * kMaxBeams + i] = (unsigned)i << 8;
      }
      wave_sync();
      continue;
    }

    // ---- 2. ' ' is a candidate: "text + pending word" of every live beam enters pyctcdecode's LM score cache (eoslog), and
    //         every pending word gets the LM score a commit would add -- once per (text, word): children that keep both
    //         inherit it.  Lane = beam: all the n-gram walks of the frame are in flight together ----
    if (use_lm && has_space) {
      for (int i0 = 0; i0 < nb; i0 += 64) {
        const int i = i0 + lane;
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
        if (put) eoslog[n_log + rank_in(pm)] = h;
        n_log += __popcll(pm);
      }
      wave_sync();
    }

    // candidates per pass: nb * cap pairs fit the merge table.  kFill / nb through the hardware reciprocal: the quotient's
    // fraction is >= 0.5 / 128 away from an integer boundary, the reciprocal is good to 1e-7 (tests replay both forms)
    const int cap = max(1, (int)(((float)kFill + 0.5f) * __builtin_amdgcn_rcpf((float)nb)));
    // survivors carried from the earlier passes of this frame: ranks lane and lane + 64
    long long c_tot[2] = {ord64(-1e300), ord64(-1e300)}, c_lgt[2] = {0, 0};
    int c_src[2] = {0, 0};
    int n_sel = 0;
    bool any_char = false;

    // table key of pair (beam bi, character c) = src -- (prefix text, last character), as the expand step forms it
    auto pair_key = [&](int sr) __attribute__((always_inline)) -> unsigned long long {
      const int bi = (sr >> 8) & 255, c = sr & 255;
      const unsigned m = S.meta[cur][bi];
      const bool grows = !(c == V || c == meta_last(m)) && !(c == space_id && meta_wlen(m) == 0);
      const unsigned long long key = S.key[cur][bi];
      return ((grows ? hmix(key, (unsigned long long)c) : key) ^ S.cmix[c]) | 1ull;
    };

    // One pass = nc candidates x nb beams, PPL pairs per lane (pair p = 64 j + lane), everything between the table and the
    // new beams in REGISTERS: a pair's slot and score, a merged prefix's combined score and logit in the lane that claimed
    // its slot.  The claimed-slot list, the pair -> slot map and the survivor records of the first version of this kernel
    // (all of them LDS round trips between dependent phases) exist no more; PPL is picked per pass from the pair count.
    auto pass = [&](auto ppl_tag, int c_lo, int nc, bool last_pass) __attribute__((always_inline)) {
      constexpr int PPL = decltype(ppl_tag)::value;
      const int npairs = nb * nc;
      const float inv_nc = __builtin_amdgcn_rcpf((float)nc);
      int slot[PPL], src[PPL];
      double score[PPL];
      unsigned claimed = 0, act = 0;
      // ---- expand: every (beam, character) pair claims / finds its slot and raises the slot's max ...  In stages over the
      //      lane's PPL pairs, so that their LDS round trips overlap: keys and home slots, then ALL first probes, then ALL
      //      claims (compare-and-swap on the slots that looked empty), and only a pair that met a foreign key walks on alone
      //      (<= 70 % load, double hashing: few do).  One pair after the other -- each with its own dependent read, swap and
      //      maximum -- took 1 500-2 500 cycles per 64 pairs, and a frame of 400 pairs ran six of those in a row ----
      unsigned long long kk[PPL];
      int stride[PPL];
      // character trie (LM built with a unigram list): the home bucket of the child's pending word, requested HERE so that
      // the table phases below hide the trip; the score step reads the answer
      ulonglong2 tfirst[PPL];
      unsigned long long wnew[PPL];
      // (branch-free: a lane whose pair index is past the end recomputes the last pair and merely takes no part in the
      // claims -- with a predicate around each pair's reads the PPL chains ran one after the other instead of together)
#pragma unroll
      for (int j = 0; j < PPL; ++j) {
        const int p = 64 * j + lane;
        if (p < npairs) act |= 1u << j;
        const int pp = min(p, npairs - 1);
        const int bi = (int)(((float)pp + 0.5f) * inv_nc);           // pp / nc (exact: tests/test_beam.py replays it)
        const int c = S.cand[c_lo + pp - bi * nc];
        const unsigned m = S.meta[cur][bi];
        const int last = meta_last(m);
        unsigned long long key = S.key[cur][bi];
        score[j] = S.logit[cur][bi] + (double)fminf(lq[c], 0.f);
        src[j] = (bi << 8) | c;
        // the prefix grows unless the character is blank, a repeat, or a space with no word pending -- one multiply,
        // no branches; the "last character" part of the key is a per-class constant (S.cmix)
        const bool grows = !(c == V || c == last) && !(c == space_id && meta_wlen(m) == 0);
        const unsigned long long kx = hmix(key, (unsigned long long)c);
        key = grows ? kx : key;
        const unsigned long long k = (key ^ S.cmix[c]) | 1ull;                          // (prefix, last char)
        // home slot and probe stride from the upper bits (bit 0 of k is forced to 1); an odd stride visits every slot
        // of the power-of-two table: double hashing, no primary clustering
        kk[j] = k;
        slot[j] = (int)((k >> 17) & (kTab - 1));
        stride[j] = (int)((k >> 40) & (kTab - 1)) | 1;
        if (trie) {                                                                     // (uniform)
          wnew[j] = hmix(S.whash[cur][bi], (unsigned long long)c);
          tfirst[j] = trie_first(lm, wnew[j]);
        }
      }
      unsigned long long seen[PPL];
#pragma unroll
      for (int j = 0; j < PPL; ++j) seen[j] = S.tkey[slot[j]];
#pragma unroll
      for (int j = 0; j < PPL; ++j)        // a slot that looked empty: try to take it (the answer says who holds it now)
        if ((act >> j & 1) && seen[j] == 0ull) {
          seen[j] = atomicCAS(&S.tkey[slot[j]], 0ull, kk[j]);
          if (seen[j] == 0ull) { claimed |= 1u << j; seen[j] = kk[j]; }
        }
#pragma unroll
      for (int j = 0; j < PPL; ++j) {
        if (act >> j & 1) {
          int i = slot[j];
          const unsigned long long k = kk[j];
          if (seen[j] != k) {              // a foreign key sits there: walk on
            while (true) {
              i = (i + stride[j]) & (kTab - 1);
              const unsigned long long e = S.tkey[i];
              if (e == k) break;
              if (e == 0) {
                const unsigned long long old = atomicCAS(&S.tkey[i], 0ull, k);
                if (old == 0ull) { claimed |= 1u << j; break; }
                if (old == k) break;
              }
            }
          }
          // The pair that CLAIMED a slot keeps its score in a register; only a pair that found its key already there -- a
          // merge: one or two per frame, against hundreds of pairs -- goes through the slot's max / sum words.  (Raising the
          // max and adding exp(score - max) for EVERY pair were two of the three LDS atomics a pair cost, and the 64-bit LDS
          // atomics are what a frame with hundreds of pairs is bound by, here as in the workgroup kernel.)
          if (!(claimed >> j & 1)) atomicMax(&S.tmx[i], ord64(score[j]));
          slot[j] = i;
        }
      }
      const bool any_merge = __ballot((act & ~claimed) != 0u) != 0ull;      // uniform
      wave_sync();
      // ---- ... a merged slot's max becomes max(claimer, contributors), the contributors add exp(score - max): hardware 2^x
      //      on a float (1 ulp), as a 2^-44 fixed-point integer -- associative, hence deterministic ----
      unsigned merged = 0;             // bit j: the slot this lane claimed for pair j has further contributors
      if (any_merge) {
#pragma unroll
        for (int j = 0; j < PPL; ++j) {
          if (claimed >> j & 1) {
            const long long mo = S.tmx[slot[j]];
            if (mo != ord64(-1e300)) { merged |= 1u << j; S.tmx[slot[j]] = max(mo, ord64(score[j])); }
          }
        }
        wave_sync();
#pragma unroll
        for (int j = 0; j < PPL; ++j) {
          if ((act & ~claimed) >> j & 1) {
            const float e = __builtin_amdgcn_exp2f((float)((score[j] - unord64(S.tmx[slot[j]])) * 1.4426950408889634));
            atomicAdd(&S.tsum[slot[j]], (unsigned long long)((double)e * kFix));
          }
        }
        wave_sync();
      }
      // ---- 3. merged prefixes, each in the lane that claimed its slot: combined score; the slot goes back to empty ----
      long long tot[PPL], lgt[PPL];
      long long my_best = max(c_tot[0], c_tot[1]);
#pragma unroll
      for (int j = 0; j < PPL; ++j) {
        // (reads unconditional -- every lane has a valid slot and parent for every j -- so that the PPL chains overlap;
        // only the writes are predicated)
        const bool mine = claimed >> j & 1;
        const int i = slot[j], bi = src[j] >> 8, c = src[j] & 255;
        float lmt = 0.f;
        if (use_lm) {
          const unsigned m = S.meta[cur][bi];
          const int last = meta_last(m), wlen = meta_wlen(m);
          const bool stay = (c == V || c == last);
          const int wlen_new = stay ? wlen : (c == space_id ? 0 : wlen + 1);
          const float commit = S.commit_lmd[cur][bi];                                   // filled by step 2 when it is needed
          // is_oov of the child's pending word: the parent's when the word stays; once outside the trie, outside for good
          bool oov = true;
          if (trie) {
            if (stay) oov = (m & kMetaOov) != 0u;
            else if (c != space_id && !(wlen > 0 && (m & kMetaOov))) oov = !trie_has_node(lm, wnew[j], tfirst[j]);
          }
          if (oov) src[j] |= kSrcOov;
          lmt = S.lm_text[cur][bi] + partial_penalty(lm.unk_offset, wlen_new, oov) + ((!stay && c == space_id && wlen > 0) ? commit : 0.f);
        }
        if (mine) S.tkey[i] = 0;
        // a prefix with a single contributor keeps that pair's score, exactly (exp(0) = 1: no logarithm)
        double logit = score[j];
        if (merged >> j & 1) {
          const double m = unord64(S.tmx[i]);
          const float e = __builtin_amdgcn_exp2f((float)((score[j] - m) * 1.4426950408889634));
          const unsigned long long s8 = S.tsum[i] + (unsigned long long)((double)e * kFix);
          S.tmx[i] = ord64(-1e300); S.tsum[i] = 0;
          logit = m + (s8 == (unsigned long long)kFix ? 0.0 : log_ge1((double)s8 * (1.0 / kFix)));
        }
        tot[j] = mine ? ord64(logit + (double)lmt) : ord64(-1e300);
        lgt[j] = __double_as_longlong(logit);
        my_best = max(my_best, tot[j]);
      }
      const long long best = wave_max_i64(my_best);
      wave_sync();
      // ---- 4. prune (max + beam_prune_logp), then the top beam_width by combined score ----
      const long long thr_prune = ord64(unord64(best) + (double)beam_prune_logp);
      const unsigned long long ubest = (unsigned long long)best ^ 0x8000000000000000ull;
      unsigned live = 0;                // bit j: entry j survives the prune; bits PPL, PPL + 1: the carried survivors
      int tot_live = 0;
#pragma unroll
      for (int j = 0; j < PPL + 2; ++j) {
        const long long tt = j < PPL ? tot[j] : c_tot[j - PPL];
        const bool lv = j < PPL ? ((claimed >> j & 1) && tt >= thr_prune) : (lane + 64 * (j - PPL) < n_sel && tt >= thr_prune);
        if (lv) live |= 1u << j;
        tot_live += __popcll(__ballot(lv));
      }
      unsigned long long prefix = 0, mask = 0;
      int want = beam_width;
      bool tie = false;                 // (uniform) the digits ran out on a bucket with more entries than wanted
      if (tot_live > beam_width) {
        // every live key lies between the prune threshold and the best score: the leading BITS those two have in common
        // (sign, exponent, the top of the mantissa) are common to all of them and cost no pass -- the first digit starts at the
        // first bit in which they differ (round 5: byte-aligned digits wasted most of the first one, 3.1 -> 2.4 digits per select
        // at beam 100, and OR-ing the live keys' differing bits over the wavefront was two reductions for at most a bit or
        // two more), the digits then walk down in steps of eight, the last clamped to bits 7..0 (an overlap with known bits
        // is harmless: they match)
        const unsigned long long d0 = ubest ^ ((unsigned long long)thr_prune ^ 0x8000000000000000ull);
        const int lead = d0 ? __clzll((long long)d0) : 64;
        if (lead > 0) { mask = lead == 64 ? ~0ull : (~0ull << (64 - lead)); prefix = ubest & mask; }
        tie = lead == 64;                       // (a prune threshold AT the best score: whatever is live is tied)
#pragma unroll 1
        for (int shift = max(0, 56 - lead); lead < 64; shift = max(0, shift - 8)) {
          for (int i = lane; i < 256; i += 64) S.hist[i] = 0;
          wave_sync();
#pragma unroll
          for (int j = 0; j < PPL + 2; ++j) {
            const unsigned long long u = (unsigned long long)(j < PPL ? tot[j] : c_tot[j - PPL]) ^ 0x8000000000000000ull;
            if ((live >> j & 1) && (u & mask) == prefix) atomicAdd(&S.hist[(int)((u >> shift) & 255)], 1);
          }
          wave_sync();
          // the bucket holding the want-th largest key, searched from the top: lane l owns bins 255 - 4l ... 252 - 4l
          int cnt[4], mine = 0;
#pragma unroll
          for (int j = 0; j < 4; ++j) { cnt[j] = S.hist[255 - (4 * lane + j)]; mine += cnt[j]; }
          int above = wave_scan_incl(mine) - mine;
          int f_bucket = -1, f_want = 0, f_whole = 0;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if (above < want && want <= above + cnt[j]) { f_bucket = 255 - (4 * lane + j); f_want = want - above; f_whole = cnt[j] == want - above; }
            above += cnt[j];
          }
          const unsigned long long fm = __ballot(f_bucket >= 0);
          const int fl = __ffsll((long long)fm) - 1;           // exactly one lane finds it (tot_live > want >= 1)
          const int bucket = __builtin_amdgcn_readlane(f_bucket, fl);
          want = __builtin_amdgcn_readlane(f_want, fl);
          const int whole = __builtin_amdgcn_readlane(f_whole, fl);
          prefix |= (unsigned long long)bucket << shift;
          mask |= 0xFFull << shift;
          if (whole || shift == 0) { tie = !whole; break; }   // the whole bucket is taken: no need to refine further
        }
        // ---- an exact tie at the cut: more entries share all 64 bits of the cut score than fit.  The ORDER of the entries is
        //      not the same here and in beam_group.hip (passes of 358 pairs here, 716 there; there the pair that claims a merged
        //      prefix is whoever came first), so position must not decide who stays: the `want` tied entries with the LARGEST
        //      table keys do -- (prefix text, last character), unique per entry and the same in every schedule.  The same digit
        //      search over the keys of the tied entries (one digit as a rule); the others then leave the live set, and what
        //      follows sees exactly `want` entries equal to the cut.  Out of line: flat synthetic posteriors reach this
        //      (tools/soak_beam.py), a model's do not ----
        if (tie) {
          const int want_tied = want;
          unsigned long long tk[PPL + 2], kprefix = 0, kmask = 0;
          unsigned tied = 0;
#pragma unroll
          for (int j = 0; j < PPL + 2; ++j) {
            tk[j] = pair_key(j < PPL ? src[j] : c_src[j - PPL]);
            const unsigned long long u = (unsigned long long)(j < PPL ? tot[j] : c_tot[j - PPL]) ^ 0x8000000000000000ull;
            if ((live >> j & 1) && u == prefix) tied |= 1u << j;
          }
#pragma unroll 1
          for (int shift = 56;; shift -= 8) {
            for (int i = lane; i < 256; i += 64) S.hist[i] = 0;
            wave_sync();
#pragma unroll
            for (int j = 0; j < PPL + 2; ++j)
              if ((tied >> j & 1) && (tk[j] & kmask) == kprefix) atomicAdd(&S.hist[(int)((tk[j] >> shift) & 255)], 1);
            wave_sync();
            int cnt[4], mine = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) { cnt[j] = S.hist[255 - (4 * lane + j)]; mine += cnt[j]; }
            int above = wave_scan_incl(mine) - mine;
            int f_bucket = -1, f_want = 0, f_whole = 0;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              if (above < want && want <= above + cnt[j]) { f_bucket = 255 - (4 * lane + j); f_want = want - above; f_whole = cnt[j] == want - above; }
              above += cnt[j];
            }
            const unsigned long long fm = __ballot(f_bucket >= 0);
            const int fl = __ffsll((long long)fm) - 1;         // exactly one lane finds it (more tied entries than wanted, want >= 1)
            const int bucket = __builtin_amdgcn_readlane(f_bucket, fl);
            want = __builtin_amdgcn_readlane(f_want, fl);
            const int whole = __builtin_amdgcn_readlane(f_whole, fl);
            kprefix |= (unsigned long long)bucket << shift;
            kmask |= 0xFFull << shift;
            if (whole || shift == 0) break;
          }
#pragma unroll
          for (int j = 0; j < PPL + 2; ++j)
            if ((tied >> j & 1) && (tk[j] & kmask) < kprefix) live &= ~(1u << j);
          want = want_tied;
        }
      }
      // selected: live and key > threshold prefix, plus the first `want` (pairs in order, then the carried survivors by
      // rank) equal to it.  A selected entry's record (pair, merged logit) goes to the sel_* rows at its rank: after the
      // last pass the new beams are built from them, otherwise they come back as carried survivors (rank = lane, lane + 64).
      int n_out = 0, eq_seen = 0;
#pragma unroll
      for (int j = 0; j < PPL + 2; ++j) {
        const long long tt = j < PPL ? tot[j] : c_tot[j - PPL];
        bool gt = false, eq = false;
        if (live >> j & 1) {
          const unsigned long long u = ((unsigned long long)tt ^ 0x8000000000000000ull) & mask;
          if (mask == 0 || u > prefix) gt = true; else if (u == prefix) eq = true;
        }
        const unsigned long long em = __ballot(eq);
        const bool take = gt || (eq && eq_seen + rank_in(em) < want);
        eq_seen += __popcll(em);
        const unsigned long long tm = __ballot(take);
        const int dst = n_out + rank_in(tm);
        if (take && dst < kMaxBeams) {
          const int sr = j < PPL ? src[j] : c_src[j - PPL];
          const long long lg = j < PPL ? lgt[j] : c_lgt[j - PPL];
          S.sel_src[dst] = sr; S.sel_lgt[dst] = lg;
          if (!last_pass) S.sel_tot[dst] = tt;
        }
        n_out += __popcll(tm);
      }
      n_sel = min(n_out, kMaxBeams);
      wave_sync();
      if (!last_pass) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int r = lane + 64 * j;
          if (r < n_sel) { c_tot[j] = S.sel_tot[r]; c_lgt[j] = S.sel_lgt[r]; c_src[j] = S.sel_src[r]; }
          else { c_tot[j] = ord64(-1e300); c_lgt[j] = 0; c_src[j] = 0; }
        }
        wave_sync();
      }
    };

#pragma unroll 1
    for (int c_lo = 0; c_lo < nc_all; c_lo += cap) {
      const int nc = min(cap, nc_all - c_lo);
      const bool last_pass = c_lo + cap >= nc_all;
      const int npairs = nb * nc;
      // PPL = ceil(pairs / 64): the cost of a pass grows with PPL (masked-off pair slots still take their issue slots), so every
      // value gets its own instantiation -- measured against the set {1, 2, 4, 6}: -3.5 % at 262 pairs per frame, -2.4 % at 140;
      // {2, 6} and {6} alone had been 4 % and 20 % slower than {1, 2, 4, 6} (the kernel's code size is not what limits it)
      switch ((npairs + 63) >> 6) {
        case 1: pass(std::integral_constant<int, 1>{}, c_lo, nc, last_pass); break;
        case 2: pass(std::integral_constant<int, 2>{}, c_lo, nc, last_pass); break;
        case 3: pass(std::integral_constant<int, 3>{}, c_lo, nc, last_pass); break;
        case 4: pass(std::integral_constant<int, 4>{}, c_lo, nc, last_pass); break;
        case 5: pass(std::integral_constant<int, 5>{}, c_lo, nc, last_pass); break;
        default: pass(std::integral_constant<int, 6>{}, c_lo, nc, last_pass); break;
      }
    }
    // ---- 5. the new beams, one per rank ----
#pragma unroll 1
    for (int r = lane; r < n_sel; r += 64) build_child(t, r, S.sel_src[r], S.sel_lgt[r], has_space, any_char);
    all_blank = __ballot(any_char) == 0ull;
    nb = n_sel;
    cur ^= 1;
    wave_sync();
  }

  // ---- final: commit pending words (LM score with </s>), merge identical texts, pick the best ----
  // Is "text + pending word" in pyctcdecode's LM cache (then its cached score, WITHOUT </s>, is what the final pass
  // uses)?  Known for beams whose own lineage put it there (`cached`); the others look their hash up in eoslog: their
  // hashes go into the (idle, empty) merge table, the wavefront walks the log and marks the hashes it meets.
  int in_cache[2] = {0, 0};
  if (use_lm) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the log's stores have reached L2 (read back past the L1 below)
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
            int q = (int)((k >> 17) & (kTab - 1));
            while (true) {
              const unsigned long long old = atomicCAS(&S.tkey[q], 0ull, k);
              if (old == 0ull || old == k) break;
              q = (q + 1) & (kTab - 1);
            }
            myslot[j] = q;
          }
        }
      }
    }
    for (int i = lane; i < kTab; i += 64) S.tsrc[i] = 0;
    wave_sync();
    for (int q = lane; q < n_log; q += 64) {
      const unsigned long long k = __hip_atomic_load(&eoslog[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      for (int i = (int)((k >> 17) & (kTab - 1));; i = (i + 1) & (kTab - 1)) {
        const unsigned long long e = S.tkey[i];
        if (e == k) { S.tsrc[i] = 1; break; }
        if (e == 0) break;
      }
    }
    wave_sync();
#pragma unroll
    for (int j = 0; j < 2; ++j) if (myslot[j] >= 0) in_cache[j] = S.tsrc[myslot[j]];
    wave_sync();
  }
  // per beam: combined final score, final text key, last-frame combined score (pyctcdecode keeps its beams sorted by it)
  double* fin = S.fin;                                                     // [kMaxBeams]
  unsigned long long* fkey = reinterpret_cast<unsigned long long*>(S.sel_lgt);   // [kMaxBeams]
  double* frank = reinterpret_cast<double*>(S.sel_tot);                    // [kMaxBeams]
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
          for (int q = 0; q < kMaxCtx; ++q) ctx[q] = S.ctx[cur][i][q];
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
  // Merge by text: log-sum-exp of the LOGIT scores, as pyctcdecode does.  "abc" with the word still pending and "abc "
  // with it committed are the same final text but not the same LM part (only the pending word is scored with </s>):
  // pyctcdecode's _merge_beams overwrites the group's entry with every further member it meets while walking its
  // score-sorted beam list, so the member with the LOWEST last-frame score provides the LM part.  Every lane takes the
  // groups whose first member it owns; the best group is the first maximum in beam order.
  // (Exact ties -- of the last-frame scores inside a group, of the groups' merged scores -- go to the larger key, not to the
  // earlier beam: the beams' order is not the same in the two kernels, see the select.)
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
  // lowest beam index among the lanes that hold the maximum (a lane's own groups are already in beam order)
  int bi_best = 0x7fffffff;
  for (unsigned long long q = wm; q; q &= q - 1) bi_best = min(bi_best, __builtin_amdgcn_readlane(my_first, __ffsll((long long)q) - 1));
  const double bs = unord64(sbest);

  // ---- trace back: the back-pointer rows come through LDS kTbRows at a time (one batch = one contiguous 6 KB read), the
  //      batch after the current one already requested while the current one is walked; the characters are collected in
  //      LDS and leave as one coalesced write (the workgroup kernel walked 501 dependent HBM round trips and reversed the
  //      text in HBM with one thread: 0.2 ms of a 3 ms search) ----
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wavefront's own back-pointer stores have reached L2
  unsigned int* rows = reinterpret_cast<unsigned int*>(S.tkey);            // [2][kTbRows][kMaxBeams], aliases tkey + tmx + tsum
  unsigned short* chars = reinterpret_cast<unsigned short*>(&S.key[0][0]);  // [kChars], aliases the beam keys / hashes / logits
  int32_t* out = out_ids + (int64_t)b * frames_ld;
  const bool in_lds = frames <= kChars;
  int n = 0, cur_b = bi_best;
  bool lead = true;                                          // still inside the trailing whitespace of the text
  constexpr int kRowRegs = kTbRows * kMaxBeams / 4 / 64;     // uint4 per lane and batch
  uint4 rr[kRowRegs];
  const int nbatch = (frames + kTbRows - 1) / kTbRows;       // batch j: frames (frames - (j + 1) kTbRows, frames - j kTbRows]
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
        if (!(lead && id == space_id)) {                      // normalise trailing whitespace
          lead = false;
          if (in_lds) chars[n] = (unsigned short)id;
          else if (lane == 0) out[frames_ld - 1 - n] = id;    // long transcripts: filled from the back, moved below
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
    // out[j] = out[frames_ld - n + j]: destination indices lie below the source indices and a chunk's loads complete
    // before its stores, so overlapping ranges are safe
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
    out_len[b] = n;
    out_score[b] = (float)bs;
  }
}

}  // namespace

size_t beam_wave_lds_bytes() { return sizeof(WaveLds); }

// utterances per workgroup (= per compute unit): a lone utterance gets a workgroup of its own; a batch is packed four to a
// compute unit so that the search of batch k leaves the rest of the chip to the acoustic pass of batch k + 1
int beam_wave_utts_per_workgroup(int batch) {
  return batch >= 4 ? 4 : (batch >= 2 ? 2 : 1);
}

int launch_beam_search_wave(const float* logp, int batch, int frames, int V1, int space_id, int beam_width,
                            float token_min_logp, float beam_prune_logp, const BeamLm* lm, unsigned int* bp,
                            int32_t* out_ids, int32_t* out_len, float* out_score, hipStream_t st,
                            const int32_t* row_frames) {
  unsigned long long* eoslog = reinterpret_cast<unsigned long long*>(bp + (size_t)batch * frames * kMaxBeams);
  const LmView v = make_lm_view(lm);
  const int use_lm = lm ? 1 : 0;
  const int upw = beam_wave_utts_per_workgroup(batch);
  const size_t lds = sizeof(WaveLds) * (size_t)upw;
  static std::atomic<uint64_t> lds_opted{0};   // per device (dyn_lds_opt_in)
  const hipError_t attr = dyn_lds_opt_in(reinterpret_cast<const void*>(beam_wave_kernel), (int)(sizeof(WaveLds) * 4), lds_opted);
  if (attr != hipSuccess) return (int)attr;
  hipLaunchKernelGGL(beam_wave_kernel, dim3((batch + upw - 1) / upw), dim3(64 * upw), lds, st, logp, batch, frames,
                     row_frames, V1, space_id, beam_width, token_min_logp, beam_prune_logp, v, use_lm, bp, eoslog, out_ids,
                     out_len, out_score);
  return 0;
}

unsigned long long beam_hash_step(unsigned long long h, unsigned long long v) { return beam_detail::hmix(h, v); }
unsigned long long beam_hash_init() { return beam_detail::kFnvOffset; }

}  // namespace vasr
