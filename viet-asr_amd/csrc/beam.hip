// CTC prefix beam search with optional back-off n-gram LM, on the device (gfx950).
//
// Replaces BeamSearchDecoderWithLM.forward (reference nemo/collections/asr/beam_search_decoder.py:95-102), which
// hands exp(log_probs[0]) to pyctcdecode on the host, one utterance at a time.  pyctcdecode / kenlm are third-party
// and absent (parity unpinned); the algorithm restated here is oracle/beam_oracle.py (file header there).
//
// One workgroup (256 threads) per utterance walks the frames; per frame
//   1. candidate characters {c : logp >= token_min_logp} U {argmax}, capped so the merge table stays < 70 % full;
//   2. every (beam, character) pair is hashed -- key = hash(prefix string incl. committed spaces, last character) --
//      into an LDS open-addressing table: identical prefixes MERGE by log-sum-exp (fp64 max via ordered-int
//      atomicMax, then a 2^-44 fixed-point atomicAdd of exp(score - max): associative, hence deterministic);
//   3. a word committed by ' ' is scored with the n-gram LM (hashed tables in HBM, back-off walk), partial words get
//      pyctcdecode's OOV penalty; beams below max-10 are dropped; the top `beam_width` are kept by a 4-pass radix
//      select on the ordered bits of the combined score (no sort);
//   4. survivors are compacted (ballot/prefix scan) and a back-pointer row is written for the final trace-back.
#include "vasr_internal.h"

namespace vasr {

namespace {

constexpr int kMaxBeams = 128;
constexpr int kSlots = 2048;
constexpr int kMaxFill = 1434;  // 70 % of kSlots
constexpr int kMaxClasses = 128;
constexpr int kMaxCtx = 4;      // LM order <= 5
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

struct Beam {
  unsigned long long key;    // hash of the prefix characters, committed spaces included
  unsigned long long whash;  // rolling hash of the current partial word (label ids)
  double logit;
  float lm_text;             // LM score of the committed words
  int last;                  // last emitted class (blank = V, none = -1)
  int wlen;                  // characters in the partial word
  int ctx[kMaxCtx];          // LM history, most recent last, -1 = empty
};

struct Slot {
  unsigned long long key;    // 0 = empty
  long long mx;              // ordered bits of the max score
  unsigned long long sum;    // fixed-point sum of exp(score - max)
  int src;                   // (beam << 8) | class
  float lm_delta;            // LM score of the word this candidate commits
  int wid;                   // id of that word (-2: nothing committed)
  int pad;
};

struct LmView {
  const unsigned long long* vkey; const int* vid; int vcap;
  const unsigned long long* nkey; const float2* nval; int ncap;
  int order, bos, eos, unk;
  float alpha, beta, unk_offset;
};

__device__ int lm_word_id(const LmView& lm, unsigned long long whash) {
  unsigned long long k = whash | 1ull;
  for (int i = (int)(k % (unsigned)lm.vcap), n = 0; n < lm.vcap; ++n, i = (i + 1 == lm.vcap ? 0 : i + 1)) {
    const unsigned long long e = lm.vkey[i];
    if (e == k) return lm.vid[i];
    if (e == 0) break;
  }
  return -1;  // out of vocabulary
}

__device__ bool lm_find(const LmView& lm, const int* ids, int n, float2* out) {
  unsigned long long k = hmix(kFnvOffset, (unsigned long long)n);
  for (int i = 0; i < n; ++i) k = hmix(k, (unsigned long long)ids[i]);
  k |= 1ull;
  for (int i = (int)(k % (unsigned)lm.ncap), c = 0; c < lm.ncap; ++c, i = (i + 1 == lm.ncap ? 0 : i + 1)) {
    const unsigned long long e = lm.nkey[i];
    if (e == k) { *out = lm.nval[i]; return true; }
    if (e == 0) break;
  }
  return false;
}

// KenLM BaseScore on a full history: log10 p(w | ctx) with back-off.
__device__ float lm_base_score(const LmView& lm, const int* ctx, int w) {
  int ids[kMaxCtx + 1];
  int n = 0;
  for (int i = 0; i < kMaxCtx; ++i)
    if (ctx[i] >= 0 && kMaxCtx - i <= lm.order - 1) ids[n++] = ctx[i];
  ids[n] = w;
  float score = 0.f;
  int start = 0;
  while (true) {
    float2 v;
    if (lm_find(lm, ids + start, n - start + 1, &v)) { score += v.x; break; }
    if (start == n) {  // unigram missing: fall back to <unk>
      int u = lm.unk;
      if (lm_find(lm, &u, 1, &v)) score += v.x; else score += -100.f;
      break;
    }
    if (lm_find(lm, ids + start, n - start, &v)) score += v.y;  // back-off weight of the context
    ++start;
  }
  return score;
}

// pyctcdecode LanguageModel.score (alpha * log10 * ln10 + beta, OOV offset, optional </s>)
__device__ float lm_word_score(const LmView& lm, const int* ctx, unsigned long long whash, bool eos, int* wid_out) {
  int wid = lm_word_id(lm, whash);
  const bool oov = wid < 0;
  if (oov) wid = lm.unk;
  float s = lm_base_score(lm, ctx, wid);
  if (oov) s += lm.unk_offset;
  if (eos) {
    int c2[kMaxCtx];
    for (int i = 0; i < kMaxCtx - 1; ++i) c2[i] = ctx[i + 1];
    c2[kMaxCtx - 1] = wid;
    s += lm_base_score(lm, c2, lm.eos);
  }
  *wid_out = wid;
  return lm.alpha * s * 2.302585092994046f + lm.beta;
}

__device__ inline float partial_penalty(float unk_offset, int wlen) {
  if (wlen <= 0) return 0.f;
  float u = unk_offset;                    // no character trie: every partial word is OOV (is_oov = 1.0)
  if (wlen > 6) u = u * (float)wlen / 6.0f;
  return u;
}

__device__ inline int block_scan_excl(int v, int* scratch, int* total) {
  // 256 threads, 4 wavefronts
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int x = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) { int y = __shfl_up(x, o, 64); if (lane >= o) x += y; }
  if (lane == 63) scratch[wave] = x;
  __syncthreads();
  int base = 0;
  for (int w = 0; w < wave; ++w) base += scratch[w];
  *total = scratch[0] + scratch[1] + scratch[2] + scratch[3];
  __syncthreads();
  return base + x - v;
}

// grid (B), block 256
__global__ __launch_bounds__(256) void beam_search_kernel(const float* __restrict__ logp, int frames, int V1,
                                                          int space_id, int beam_width, float token_min_logp,
                                                          float beam_prune_logp, LmView lm, int use_lm,
                                                          unsigned int* __restrict__ bp_all,
                                                          int32_t* __restrict__ out_ids, int32_t* __restrict__ out_len,
                                                          float* __restrict__ out_score) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  Slot* slots = reinterpret_cast<Slot*>(smem);
  Beam* beams = reinterpret_cast<Beam*>(slots + kSlots);
  Beam* nbeams = beams + kMaxBeams;
  double* lp = reinterpret_cast<double*>(nbeams + kMaxBeams);  // [kMaxClasses]
  int* cand = reinterpret_cast<int*>(lp + kMaxClasses);        // [kMaxClasses]
  int* hist = cand + kMaxClasses;                              // [256]
  int* misc = hist + 256;                                      // [16]
  long long* best = reinterpret_cast<long long*>(misc + 16);   // [1]

  const int tid = threadIdx.x, b = blockIdx.x, V = V1 - 1;
  const float* lrow = logp + (int64_t)b * frames * V1;
  unsigned int* bp = bp_all + (int64_t)b * frames * kMaxBeams;

  if (tid == 0) {
    Beam s{};
    s.key = kFnvOffset; s.whash = kFnvOffset; s.logit = 0.0; s.lm_text = 0.f; s.last = -1; s.wlen = 0;
    for (int i = 0; i < kMaxCtx; ++i) s.ctx[i] = -1;
    if (use_lm) s.ctx[kMaxCtx - 1] = lm.bos;
    beams[0] = s;
    misc[0] = 1;  // live beams
  }
  __syncthreads();

  for (int t = 0; t < frames; ++t) {
    const int nb = misc[0];
    // ---- 1. log-probs (pyctcdecode: log(clip(p, 1e-15, 1))) and candidate characters ----
    if (tid < V1) lp[tid] = log(fmin(fmax(exp((double)lrow[(int64_t)t * V1 + tid]), 1e-15), 1.0));
    if (tid == 0) { misc[1] = 0; *best = ord64(-1e300); }
    for (int i = tid; i < kSlots; i += 256) { slots[i].key = 0; slots[i].mx = ord64(-1e300); slots[i].sum = 0; }
    __syncthreads();
    if (tid < V1) {
      const double v = lp[tid];
      int rank = 0;            // number of classes strictly better (ties: lower index first) -> argmax has rank 0
      for (int j = 0; j < V1; ++j) rank += (lp[j] > v) || (lp[j] == v && j < tid);
      const int cap = max(1, kMaxFill / nb);
      if ((v >= (double)token_min_logp || rank == 0) && rank < cap) cand[atomicAdd(&misc[1], 1)] = tid;
    }
    __syncthreads();
    const int nc = misc[1];
    // ---- 2. expand: phase 1 claims a slot and raises its max, phase 2 adds exp(score - max) ----
    for (int phase = 0; phase < 2; ++phase) {
      for (int p = tid; p < nb * nc; p += 256) {
        const int bi = p / nc, c = cand[p % nc];
        const Beam& s = beams[bi];
        unsigned long long key = s.key;
        if (!(c == V || c == s.last)) {
          if (c == space_id) { if (s.wlen > 0) key = hmix(key, (unsigned long long)c); }
          else key = hmix(key, (unsigned long long)c);
        }
        unsigned long long k = hmix(key, (unsigned long long)(c + 7)) | 1ull;   // (prefix, last char)
        const double score = s.logit + lp[c];
        int i = (int)(k & (kSlots - 1));
        while (true) {
          const unsigned long long e = slots[i].key;
          if (e == k) break;
          if (e == 0) {
            const unsigned long long old = atomicCAS(&slots[i].key, 0ull, k);
            if (old == 0ull) { slots[i].src = (bi << 8) | c; break; }
            if (old == k) break;
          }
          i = (i + 1) & (kSlots - 1);
        }
        if (phase == 0) atomicMax(&slots[i].mx, ord64(score));
        else atomicAdd(&slots[i].sum, (unsigned long long)(exp(score - unord64(slots[i].mx)) * kFix));
      }
      __syncthreads();
    }
    // ---- 3. LM scoring of committed words, combined score, running max ----
    for (int i = tid; i < kSlots; i += 256) {
      Slot& sl = slots[i];
      if (sl.key == 0) continue;
      const int bi = sl.src >> 8, c = sl.src & 255;
      const Beam& s = beams[bi];
      const double logit = unord64(sl.mx) + log((double)sl.sum / kFix);
      const bool stay = (c == V || c == s.last);
      const bool commit = !stay && c == space_id && s.wlen > 0;
      int wlen_new = stay ? s.wlen : (c == space_id ? 0 : s.wlen + 1);
      float lm_total = 0.f;
      sl.wid = -2; sl.lm_delta = 0.f;
      if (use_lm) {
        if (commit) sl.lm_delta = lm_word_score(lm, s.ctx, s.whash, false, &sl.wid);
        lm_total = s.lm_text + sl.lm_delta + partial_penalty(lm.unk_offset, wlen_new);
      }
      const double total = logit + (double)lm_total;
      sl.mx = ord64(total);                 // reuse: ordered combined score
      sl.sum = (unsigned long long)__double_as_longlong(logit);
      atomicMax(best, ord64(total));
    }
    __syncthreads();
    // ---- 4. prune (max + beam_prune_logp) and radix-select the top beam_width by combined score ----
    const long long thr_prune = ord64(unord64(*best) + (double)beam_prune_logp);
    unsigned long long prefix = 0, mask = 0;
    int want = beam_width;   // how many still to take among keys matching the prefix
    {
      int cnt = 0;
      for (int i = tid; i < kSlots; i += 256) cnt += (slots[i].key != 0 && slots[i].mx >= thr_prune);
      int tot;
      block_scan_excl(cnt, misc + 4, &tot);
      if (tot > beam_width) {
        for (int shift = 56; shift >= 0; shift -= 8) {
          hist[tid] = 0;
          __syncthreads();
          for (int i = tid; i < kSlots; i += 256) {
            if (slots[i].key == 0 || slots[i].mx < thr_prune) continue;
            const unsigned long long u = (unsigned long long)slots[i].mx ^ 0x8000000000000000ull;
            if ((u & mask) == prefix) atomicAdd(&hist[(int)((u >> shift) & 255)], 1);
          }
          __syncthreads();
          if (tid == 0) {
            int acc = 0, d = 255;
            for (; d >= 0; --d) { if (acc + hist[d] >= want) break; acc += hist[d]; }
            misc[2] = d; misc[3] = want - acc;
          }
          __syncthreads();
          prefix |= (unsigned long long)misc[2] << shift;
          mask |= 0xFFull << shift;
          want = misc[3];
          __syncthreads();
        }
      } else { prefix = 0; mask = 0; want = beam_width; }
    }
    // selected: score > threshold key, plus the first `want` (in slot order) equal to it
    int sel_gt = 0, sel_eq = 0;
    for (int i = tid; i < kSlots; i += 256) {
      if (slots[i].key == 0 || slots[i].mx < thr_prune) continue;
      const unsigned long long u = (unsigned long long)slots[i].mx ^ 0x8000000000000000ull;
      if (mask == 0 || u > prefix) ++sel_gt; else if (u == prefix) ++sel_eq;
    }
    int tot_gt, tot_eq;
    const int off_gt = block_scan_excl(sel_gt, misc + 4, &tot_gt);
    const int off_eq = block_scan_excl(sel_eq, misc + 4, &tot_eq);
    const int take_eq = mask == 0 ? 0 : min(want, tot_eq);
    {
      int ig = off_gt, ie = off_eq;
      for (int i = tid; i < kSlots; i += 256) {
        if (slots[i].key == 0 || slots[i].mx < thr_prune) continue;
        const unsigned long long u = (unsigned long long)slots[i].mx ^ 0x8000000000000000ull;
        int dst = -1;
        if (mask == 0 || u > prefix) dst = ig++;
        else if (u == prefix) { if (ie < take_eq) dst = tot_gt + ie; ++ie; }
        if (dst < 0 || dst >= kMaxBeams) continue;
        const Slot& sl = slots[i];
        const int bi = sl.src >> 8, c = sl.src & 255;
        const Beam& s = beams[bi];
        Beam n = s;
        const bool stay = (c == V || c == s.last);
        unsigned int appended = 0;
        if (!stay) {
          if (c == space_id) {
            if (s.wlen > 0) {
              n.key = hmix(s.key, (unsigned long long)c);
              appended = c + 1;
              n.lm_text = s.lm_text + sl.lm_delta;
              if (use_lm) { for (int q = 0; q < kMaxCtx - 1; ++q) n.ctx[q] = s.ctx[q + 1]; n.ctx[kMaxCtx - 1] = sl.wid; }
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
        n.logit = __longlong_as_double((long long)sl.sum);
        nbeams[dst] = n;
        bp[(int64_t)t * kMaxBeams + dst] = ((unsigned)bi << 8) | appended;
      }
    }
    __syncthreads();
    if (tid == 0) misc[0] = min(kMaxBeams, tot_gt + take_eq);
    for (int i = tid; i < min(kMaxBeams, tot_gt + take_eq); i += 256) beams[i] = nbeams[i];
    __syncthreads();
  }

  // ---- final: commit pending words (LM score with </s>), merge identical texts, pick the best ----
  const int nb = misc[0];
  double* fin = lp;  // [kMaxBeams] combined score per beam
  unsigned long long* fkey = reinterpret_cast<unsigned long long*>(slots);  // [kMaxBeams]
  if (tid < nb) {
    const Beam& s = beams[tid];
    double total = s.logit;
    if (use_lm) {
      float lmv = s.lm_text;
      int wid;
      if (s.wlen > 0) lmv += lm_word_score(lm, s.ctx, s.whash, true, &wid);
      total += (double)lmv;
    }
    fin[tid] = total;
    fkey[tid] = s.wlen > 0 ? hmix(s.key, (unsigned long long)space_id) : s.key;
  }
  __syncthreads();
  if (tid == 0) {
    // merge by text: log-sum-exp of the LOGIT scores is what pyctcdecode does; the LM part is per text.  Beams with the
    // same final text share the LM score, so combine through the logit difference.
    int bi = 0;
    double bs = -1e300;
    for (int i = 0; i < nb; ++i) {
      bool first = true;
      for (int j = 0; j < i; ++j) if (fkey[j] == fkey[i]) { first = false; break; }
      if (!first) continue;
      double m = beams[i].logit;
      for (int j = i + 1; j < nb; ++j) if (fkey[j] == fkey[i]) m = fmax(m, beams[j].logit);
      double ssum = 0;
      for (int j = i; j < nb; ++j) if (fkey[j] == fkey[i]) ssum += exp(beams[j].logit - m);
      const double merged = (fin[i] - beams[i].logit) + m + log(ssum);
      if (merged > bs) { bs = merged; bi = i; }
    }
    // trace back
    int n = 0, cur = bi;
    int32_t* out = out_ids + (int64_t)b * frames;
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

size_t beam_lds_bytes() {
  return sizeof(Slot) * kSlots + sizeof(Beam) * 2 * kMaxBeams + sizeof(double) * kMaxClasses + sizeof(int) * (kMaxClasses + 256 + 16) + 16;
}

void launch_beam_search(const float* logp, int batch, int frames, int V1, int space_id, int beam_width,
                        float token_min_logp, float beam_prune_logp, const BeamLm* lm, unsigned int* bp,
                        int32_t* out_ids, int32_t* out_len, float* out_score, hipStream_t st) {
  LmView v{};
  int use_lm = 0;
  if (lm) {
    use_lm = 1;
    v.vkey = lm->vkey; v.vid = lm->vid; v.vcap = lm->vcap; v.nkey = lm->nkey;
    v.nval = reinterpret_cast<const float2*>(lm->nval); v.ncap = lm->ncap; v.order = lm->order; v.bos = lm->bos;
    v.eos = lm->eos; v.unk = lm->unk; v.alpha = lm->alpha; v.beta = lm->beta; v.unk_offset = lm->unk_offset;
  }
  const size_t lds = beam_lds_bytes();
  static bool once = [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(beam_search_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)beam_lds_bytes());
    return true;
  }();
  (void)once;
  hipLaunchKernelGGL(beam_search_kernel, dim3(batch), dim3(256), lds, st, logp, frames, V1, space_id, beam_width,
                     token_min_logp, beam_prune_logp, v, use_lm, bp, out_ids, out_len, out_score);
}

unsigned long long beam_hash_step(unsigned long long h, unsigned long long v) { return hmix(h, v); }
unsigned long long beam_hash_init() { return kFnvOffset; }

}  // namespace vasr
