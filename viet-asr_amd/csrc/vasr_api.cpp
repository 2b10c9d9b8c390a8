// C ABI of libvasr_hip.so (see include/vasr.h): handle, weight intake, BN folding / packing,
// workspace planning and the launch sequences for each stage of the path.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <algorithm>
#include <vector>

#include "vasr.h"
#include "vasr_devtools.h"
#include "vasr_internal.h"

using namespace vasr;

namespace vasr { thread_local LaunchProbe g_probe; }

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

#define HIP_TRY(expr)                                                                          \
  do {                                                                                         \
    hipError_t e_ = (expr);                                                                    \
    if (e_ != hipSuccess) return fail(VASR_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
  } while (0)

struct HostTensor {
  std::vector<float> data;
  std::vector<int64_t> shape;
};

struct ConvLayer {
  int cin = 0, cout = 0, kernel = 1, stride = 1, dilation = 1, pad = 0;
  int m_pad = 0;             // pointwise: rows of the packed weight (multiple of 128)
  float* d_w = nullptr;      // depthwise [C][K]; pointwise: MFMA A-fragment order (fp32)
  unsigned short* d_w3 = nullptr;  // pointwise: 3 x bf16 split fragments (encoder_pw_split.hip), when the shape allows
  unsigned short* d_w16 = nullptr; // pointwise: 2 x fp16 scaled split fragments, same condition
  float w16_inv = 1.f;             // 1 / (power-of-two scale of the fp16 pack)
  unsigned int* d_taps = nullptr;  // depthwise, Toeplitz / MFMA form: [C][tap_tsz] (hi | lo << 16) fp16 tap tables
  float* d_tap_inv = nullptr;      // [C] 1 / (power-of-two scale of the channel's taps)
  int tap_tsz = 0;
  float* d_ftaps = nullptr;        // depthwise, fused dw -> pw kernel (encoder_fused.hip): [C / 2][taps per pair][2]
  float f_l1 = 0.f;                // max_c sum_k |w[c][k]|: bound of the depthwise output per unit of input
  float* d_scale = nullptr;  // [m_pad]
  float* d_shift = nullptr;  // [m_pad]
  int step = -1;             // index in the MaskedConv1d length chain
};

struct SubBlock {
  bool separable = true;
  ConvLayer dw, pw;
};

struct Block {
  vasr_block_desc d;
  std::vector<SubBlock> subs;
  bool has_res = false;
  ConvLayer res;
  bool fused_res = false;   // residual 1x1 conv folded into the last sub-block's GEMM (dual-source K)
  ConvLayer fused;          // weights [s1*W1 | s2*W2], scale 1, shift h1 + h2; cin = K1 + K2
  int fused_k1 = 0;
  int first_step = 0;
};

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
// Default 3 (2 x fp16 scaled split): against fp64 its error is not larger than mode 0's or mode 1's (K = 256 ... 1024,
// Gaussian / ReLU'd / 30-octave inputs), every reference fixture holds with the same tolerance, results stay independent
// of batch composition, at half the matrix work of mode 1 (QuartzNet15x5, 64 x 10 s: GEMMs 5.4 -> 3.7 ms per batch).
constexpr int kDefaultGemmMode = 3;
int parse_gemm_mode(const char* s) {
  if (!s) return kDefaultGemmMode;
  if (!strcmp(s, "fp32")) return 0;
  if (!strcmp(s, "bf16x3")) return 1;
  if (!strcmp(s, "bf16x2")) return 2;
  if (!strcmp(s, "f16x2")) return 3;
  return kDefaultGemmMode;
}
constexpr int kMaxSlices = 4;
constexpr int kAmaxTabs = 4;

}  // namespace

struct vasr_handle {
  bool has_frontend = false, has_encoder = false, has_decoder = false, finalized = false;
  vasr_frontend_desc fe{};
  std::vector<float> fe_window, fe_fb;
  int feat_in = 0, dec_feat_in = 0, num_classes = 0;
  std::vector<Block> blocks;
  std::vector<LenStep> steps;
  std::map<std::string, HostTensor> weights;
  std::vector<void*> dev_allocs;
  // device tables
  FrontendTables ft{};
  LenStep* d_steps = nullptr;
  ConvLayer dec;
  int c_mid_max = 0, c_last = 0;
  // optional per-kernel-class HIP-event timing (vasr_profile_begin/end)
  // batch slicing across internal streams (vasr_set_slices)
  // measured on MI355X (QuartzNet15x5, B=64, 512x128 GEMM tiles pinned): 1 slice 7.35 ms, 2 slices 7.41 ms, 2 slices
  // phase-shifted by 40 / 100 / 300 us 7.39 / 7.34 / 7.68 ms -- the kernels of the two streams do not overlap in any
  // useful way (one workgroup per CU each), so slicing is OFF by default
  int slices = getenv("VASR_SLICES") ? atoi(getenv("VASR_SLICES")) : 1;
  bool slice_ready = false;
  bool row_independent = false;   // vasr_set_row_independent
  int busy_cus = 0;               // vasr_set_busy_cus
  DevSwitches sw = dev_switches();   // kernel-selection switches of the devtools build (vasr_internal.h); defaults in the product
  int n_cu = 256;                 // compute units of the device the handle was finalized on (tile-fill decisions)
  hipStream_t slice_stream[kMaxSlices] = {};
  hipEvent_t slice_done[kMaxSlices] = {}, slice_fork{};
  // 0 = v_mfma_f32_32x32x2_f32 (exact fp32 fmaf chain), 1 = 3 x bf16 split operands on v_mfma_f32_32x32x16_bf16 (measured
  // max error against fp64 slightly LOWER than mode 0's: 3.6e-6 vs 4.7e-6 at K = 512), 2 = reduced 2 x bf16 (opt-in),
  // 3 = 2 x fp16 scaled split operands on v_mfma_f32_32x32x16_f16 (half the matrix work of mode 1, see vasr.h)
  int gemm_mode = parse_gemm_mode(getenv("VASR_GEMM"));
  bool profiling = false;
  struct ProfRec { hipEvent_t a, b; int cls; double flops, bytes; };
  std::vector<ProfRec> prof;
  std::vector<hipEvent_t> ev_pool;
};

#ifdef VASR_DEVTOOLS
// The devtools build's kernel-selection switches (vasr_internal.h DevSwitches): the environment is read HERE, once per
// process, and nowhere else; a value outside a switch's documented set aborts instead of silently meaning "default".
const vasr::DevSwitches& vasr::dev_switches() {
  static const DevSwitches sw = [] {
    DevSwitches s;
    auto num = [](const char* name, int dflt, std::initializer_list<int> allowed) {
      const char* e = getenv(name);
      if (!e) return dflt;
      const int v = atoi(e);
      for (int a : allowed) if (a == v) return v;
      fprintf(stderr, "vasr (devtools build): %s=%s is not a documented value -- refusing to guess\n", name, e);
      abort();
      return dflt;   // (not reached)
    };
    s.pw3_tile = num("VASR_PW3_TILE", 0, {0, 1, 2, 3, 4, 5});
    s.pw_lat = num("VASR_PW_LAT", 1, {0, 1});
    s.dw_pair = num("VASR_DW_PAIR", 1, {0, 1}) != 0;
    s.dw_mfma = num("VASR_DW_MFMA", 1, {0, 1}) != 0;
    s.dw_upw = num("VASR_DW_UPW", 0, {0, 1, 2, 3, 4, 5, 6, 7, 8});
    s.fused = num("VASR_FUSED", 1, {0, 1}) != 0;
    s.fused_min_tiles = getenv("VASR_FUSED_MIN_TILES") ? atoi(getenv("VASR_FUSED_MIN_TILES")) : 0;
    s.fused_tile = num("VASR_FUSED_TILE", 0, {0, 64, 128});
    s.fused_residual = num("VASR_NO_FUSED_RESIDUAL", 0, {0, 1}) == 0;
    s.beam_group = num("VASR_BEAM_GROUP", -1, {-1, 0, 1, 4});
    return s;
  }();
  return sw;
}
#endif

struct vasr_lm {
  BeamLm view{};
  std::vector<void*> allocs;
};

namespace {

template <class T>
int upload(vasr_handle* h, const std::vector<T>& v, T** out) {
  void* p = nullptr;
  HIP_TRY(hipMalloc(&p, std::max<size_t>(v.size() * sizeof(T), 16)));
  h->dev_allocs.push_back(p);
  HIP_TRY(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  *out = static_cast<T*>(p);
  return 0;
}

int same_pad(int k, int stride, int dil, int* pad) {
  // get_same_padding (parts/jasper.py:60-65)
  if (stride > 1 && dil > 1) return fail(VASR_ERR_INVALID, "Only stride OR dilation may be greater than 1");
  *pad = dil > 1 ? (dil * k) / 2 - 1 : k / 2;
  return 0;
}

int64_t conv_out_frames(int64_t t, const ConvLayer& c) {
  return (t + 2 * c.pad - (int64_t)c.dilation * (c.kernel - 1) - 1) / c.stride + 1;
}

const HostTensor* find(const vasr_handle* h, const std::string& key) {
  auto it = h->weights.find(key);
  return it == h->weights.end() ? nullptr : &it->second;
}

int need(const vasr_handle* h, const std::string& key, size_t numel, const HostTensor** out) {
  const HostTensor* t = find(h, key);
  if (!t) return fail(VASR_ERR_STATE, "missing weight '%s'", key.c_str());
  if (t->data.size() != numel)
    return fail(VASR_ERR_INVALID, "weight '%s' has %zu elements, expected %zu", key.c_str(), t->data.size(), numel);
  *out = t;
  return 0;
}

// eval-mode BatchNorm1d(eps=1e-3) -> y = x * scale + shift  (parts/jasper.py:392)
int fold_bn(vasr_handle* h, const std::string& prefix, int c, int c_pad, ConvLayer* L) {
  const HostTensor *g, *b, *m, *v;
  int rc;
  if ((rc = need(h, prefix + ".weight", c, &g))) return rc;
  if ((rc = need(h, prefix + ".bias", c, &b))) return rc;
  if ((rc = need(h, prefix + ".running_mean", c, &m))) return rc;
  if ((rc = need(h, prefix + ".running_var", c, &v))) return rc;
  std::vector<float> sc(c_pad, 0.f), sh(c_pad, 0.f);
  for (int i = 0; i < c; ++i) {
    // same association as ATen's CPU eval path: alpha = w * invstd, beta = b - mean * alpha (fp32)
    const float invstd = 1.0f / std::sqrt(v->data[i] + 1e-3f);
    const float alpha = g->data[i] * invstd;
    sc[i] = alpha;
    sh[i] = b->data[i] - m->data[i] * alpha;
  }
  if ((rc = upload(h, sc, &L->d_scale))) return rc;
  return upload(h, sh, &L->d_shift);
}

int bn_affine(vasr_handle* h, const std::string& prefix, int c, std::vector<float>* alpha, std::vector<float>* beta) {
  const HostTensor *g, *b, *m, *v;
  int rc;
  if ((rc = need(h, prefix + ".weight", c, &g)) || (rc = need(h, prefix + ".bias", c, &b)) ||
      (rc = need(h, prefix + ".running_mean", c, &m)) || (rc = need(h, prefix + ".running_var", c, &v)))
    return rc;
  alpha->resize(c);
  beta->resize(c);
  for (int i = 0; i < c; ++i) {
    const float invstd = 1.0f / std::sqrt(v->data[i] + 1e-3f);
    (*alpha)[i] = g->data[i] * invstd;
    (*beta)[i] = b->data[i] - m->data[i] * (*alpha)[i];
  }
  return 0;
}

// Last sub-block GEMM and residual GEMM of one JasperBlock as ONE reduction over K1 + K2:
//   BN1(W1 d) + BN2(W2 x) = [a1*W1 | a2*W2] [d ; x] + (b1 + b2)        (parts/jasper.py:428-439)
int pack_fused_residual(vasr_handle* h, const std::string& w1_key, const std::string& bn1, const std::string& w2_key,
                        const std::string& bn2, int cout, int k1, int k2, ConvLayer* L) {
  const HostTensor *w1, *w2;
  std::vector<float> a1, b1, a2, b2;
  int rc;
  if ((rc = need(h, w1_key, (size_t)cout * k1, &w1)) || (rc = need(h, w2_key, (size_t)cout * k2, &w2)) ||
      (rc = bn_affine(h, bn1, cout, &a1, &b1)) || (rc = bn_affine(h, bn2, cout, &a2, &b2)))
    return rc;
  const int K = k1 + k2;
  std::vector<float> w((size_t)cout * K);
  for (int m = 0; m < cout; ++m) {
    for (int k = 0; k < k1; ++k) w[(size_t)m * K + k] = a1[m] * w1->data[(size_t)m * k1 + k];
    for (int k = 0; k < k2; ++k) w[(size_t)m * K + k1 + k] = a2[m] * w2->data[(size_t)m * k2 + k];
  }
  L->cin = K;
  L->cout = cout;
  L->m_pad = (int)align_up(cout, 128);
  std::vector<float> wt((size_t)K * L->m_pad, 0.f), sc(L->m_pad, 1.f), sh(L->m_pad, 0.f);
  pack_pointwise_weights(w.data(), cout, K, L->m_pad, wt.data());
  for (int m = 0; m < cout; ++m) sh[m] = b1[m] + b2[m];
  if (pointwise_split_supported(L->m_pad, K, k1)) {
    std::vector<unsigned short> w3((size_t)K * L->m_pad * 3);
    pack_pointwise_weights_bf16x3(w.data(), cout, K, L->m_pad, w3.data());
    if ((rc = upload(h, w3, &L->d_w3))) return rc;
    std::vector<unsigned short> w16((size_t)K * L->m_pad * 2);
    L->w16_inv = pack_pointwise_weights_f16x2(w.data(), cout, K, L->m_pad, w16.data());
    if ((rc = upload(h, w16, &L->d_w16))) return rc;
  }
  if ((rc = upload(h, wt, &L->d_w)) || (rc = upload(h, sc, &L->d_scale))) return rc;
  return upload(h, sh, &L->d_shift);
}

// [cout][cin][1] -> MFMA A-fragment order
int pack_pointwise(vasr_handle* h, const std::string& key, int cout, int cin, ConvLayer* L) {
  const HostTensor* w;
  int rc;
  if ((rc = need(h, key, (size_t)cout * cin, &w))) return rc;
  // (a K depth of 32 would select the 128 x 256 tile of the fp32 kernel, whose last time tile assumes a 256-frame pitch)
  if (cin % 64) return fail(VASR_ERR_UNSUPPORTED, "%s: in_channels %d is not a multiple of 64", key.c_str(), cin);
  L->cin = cin;
  L->cout = cout;
  L->m_pad = (int)align_up(cout, 128);
  std::vector<float> wt((size_t)cin * L->m_pad, 0.f);
  pack_pointwise_weights(w->data.data(), cout, cin, L->m_pad, wt.data());
  if (pointwise_split_supported(L->m_pad, cin, 0)) {
    std::vector<unsigned short> w3((size_t)cin * L->m_pad * 3);
    pack_pointwise_weights_bf16x3(w->data.data(), cout, cin, L->m_pad, w3.data());
    if ((rc = upload(h, w3, &L->d_w3))) return rc;
    std::vector<unsigned short> w16((size_t)cin * L->m_pad * 2);
    L->w16_inv = pack_pointwise_weights_f16x2(w->data.data(), cout, cin, L->m_pad, w16.data());
    if ((rc = upload(h, w16, &L->d_w16))) return rc;
  }
  return upload(h, wt, &L->d_w);
}

int build_frontend(vasr_handle* h) {
  const vasr_frontend_desc& fe = h->fe;
  const int nfft = fe.n_fft, nb = nfft / 2 + 1;
  std::vector<float> win(nfft, 0.f);
  const int off = (nfft - fe.win_length) / 2;  // torch.stft centres the window inside n_fft
  for (int i = 0; i < fe.win_length; ++i) win[off + i] = h->fe_window[i];
  std::vector<float> tw256(512), tw512(2 * 258, 0.f);
  for (int m = 0; m < 256; ++m) {
    tw256[2 * m] = (float)std::cos(-2.0 * M_PI * m / 256.0);
    tw256[2 * m + 1] = (float)std::sin(-2.0 * M_PI * m / 256.0);
  }
  for (int k = 0; k <= 256; ++k) {
    tw512[2 * k] = (float)std::cos(-2.0 * M_PI * k / 512.0);
    tw512[2 * k + 1] = (float)std::sin(-2.0 * M_PI * k / 512.0);
  }
  std::vector<float> mw((size_t)fe.n_mels * kMelTaps, 0.f);
  std::vector<int32_t> lo(fe.n_mels, 0);
  for (int f = 0; f < fe.n_mels; ++f) {
    const float* row = &h->fe_fb[(size_t)f * nb];
    int first = -1, last = -1;
    for (int k = 0; k < nb; ++k)
      if (row[k] != 0.f) { if (first < 0) first = k; last = k; }
    if (first < 0) { lo[f] = 0; continue; }
    if (last - first + 1 > kMelTaps)
      return fail(VASR_ERR_UNSUPPORTED, "mel filter %d spans %d bins (> %d)", f, last - first + 1, kMelTaps);
    lo[f] = first;
    for (int k = first; k <= last; ++k) mw[(size_t)f * kMelTaps + (k - first)] = row[k];
  }
  float *d_win, *d_t256, *d_t512, *d_mw;
  int32_t* d_lo;
  int rc;
  if ((rc = upload(h, win, &d_win)) || (rc = upload(h, tw256, &d_t256)) || (rc = upload(h, tw512, &d_t512)) ||
      (rc = upload(h, mw, &d_mw)) || (rc = upload(h, lo, &d_lo)))
    return rc;
  h->ft = FrontendTables{d_win, d_t256, d_t512, d_mw, d_lo, h->fe.log_guard_clamp ? 1 : 0};
  return 0;
}

int build_encoder(vasr_handle* h) {
  int cin = h->feat_in, step = 0, rc;
  h->steps.clear();
  h->c_mid_max = cin;
  for (size_t i = 0; i < h->blocks.size(); ++i) {
    Block& B = h->blocks[i];
    const vasr_block_desc& d = B.d;
    int k = d.kernel;
    if (k % 2 == 0) k += 1;  // compute_new_kernel_size (parts/jasper.py:52-57)
    int pad;
    if ((rc = same_pad(k, d.stride, d.dilation, &pad))) return rc;
    B.first_step = step;
    B.subs.resize(d.repeat);
    int c = cin, j = 0;
    char key[160];
    for (int r = 0; r < d.repeat; ++r) {
      SubBlock& S = B.subs[r];
      S.separable = d.separable != 0;
      if (S.separable) {
        const HostTensor* w;
        snprintf(key, sizeof key, "encoder.%zu.mconv.%d.conv.weight", i, j);
        if ((rc = need(h, key, (size_t)c * k, &w))) return rc;
        S.dw.cin = S.dw.cout = c;
        S.dw.kernel = k; S.dw.stride = d.stride; S.dw.dilation = d.dilation; S.dw.pad = pad;
        S.dw.step = step++;
        h->steps.push_back(LenStep{k, d.stride, d.dilation, pad});
        if ((rc = upload(h, w->data, &S.dw.d_w))) return rc;
        S.dw.tap_tsz = d.stride == 1 ? depthwise_mfma_table_size(k, d.dilation) : 0;
        if (S.dw.tap_tsz) {
          std::vector<unsigned int> tab((size_t)c * S.dw.tap_tsz);
          std::vector<float> inv(c);
          for (int ch = 0; ch < c; ++ch)
            inv[ch] = pack_depthwise_taps_f16x2(&w->data[(size_t)ch * k], k, d.dilation, S.dw.tap_tsz, &tab[(size_t)ch * S.dw.tap_tsz]);
          if ((rc = upload(h, tab, &S.dw.d_taps)) || (rc = upload(h, inv, &S.dw.d_tap_inv))) return rc;
        }
        if (fused_dwpw_supported(c, d.filters, k, d.stride, d.dilation)) {
          std::vector<float> ft((size_t)(c / 2) * fused_dwpw_taps_per_pair(k) * 2);
          S.dw.f_l1 = pack_fused_taps(w->data.data(), c, k, ft.data());
          if ((rc = upload(h, ft, &S.dw.d_ftaps))) return rc;
        }
        snprintf(key, sizeof key, "encoder.%zu.mconv.%d.conv.weight", i, j + 1);
        if ((rc = pack_pointwise(h, key, d.filters, c, &S.pw))) return rc;
        S.pw.step = step++;
        h->steps.push_back(LenStep{1, 1, 1, 0});
        snprintf(key, sizeof key, "encoder.%zu.mconv.%d", i, j + 2);
        if ((rc = fold_bn(h, key, d.filters, S.pw.m_pad, &S.pw))) return rc;
        j += 3;
      } else {
        if (k != 1 || d.stride != 1)
          return fail(VASR_ERR_UNSUPPORTED, "block %zu: non-separable conv with kernel %d is not implemented", i, k);
        snprintf(key, sizeof key, "encoder.%zu.mconv.%d.conv.weight", i, j);
        if ((rc = pack_pointwise(h, key, d.filters, c, &S.pw))) return rc;
        S.pw.step = step++;
        h->steps.push_back(LenStep{1, 1, 1, 0});
        snprintf(key, sizeof key, "encoder.%zu.mconv.%d", i, j + 1);
        if ((rc = fold_bn(h, key, d.filters, S.pw.m_pad, &S.pw))) return rc;
        j += 2;
      }
      if (r != d.repeat - 1) j += 2;  // activation + dropout entries of the ModuleList
      c = d.filters;
    }
    B.has_res = d.residual != 0;
    if (B.has_res) {
      snprintf(key, sizeof key, "encoder.%zu.res.0.0.conv.weight", i);
      if ((rc = pack_pointwise(h, key, d.filters, cin, &B.res))) return rc;
      snprintf(key, sizeof key, "encoder.%zu.res.0.1", i);
      if ((rc = fold_bn(h, key, d.filters, B.res.m_pad, &B.res))) return rc;
      // fold the residual branch into the last sub-block's GEMM when both reductions tile evenly
      const SubBlock& last = B.subs.back();
      const int k1 = last.pw.cin, k2 = cin;
      const int chunk = d.filters % 512 == 0 ? 128 : (d.filters % 256 == 0 ? 64 : 32);
      if (d.stride == 1 && k1 % chunk == 0 && k2 % chunk == 0 && h->sw.fused_residual) {
        char w1[160], bn1[160], w2[160], bn2[160];
        const int jl = j - (last.separable ? 3 : 2);
        snprintf(w1, sizeof w1, "encoder.%zu.mconv.%d.conv.weight", i, jl + (last.separable ? 1 : 0));
        snprintf(bn1, sizeof bn1, "encoder.%zu.mconv.%d", i, jl + (last.separable ? 2 : 1));
        snprintf(w2, sizeof w2, "encoder.%zu.res.0.0.conv.weight", i);
        snprintf(bn2, sizeof bn2, "encoder.%zu.res.0.1", i);
        if ((rc = pack_fused_residual(h, w1, bn1, w2, bn2, d.filters, k1, k2, &B.fused))) return rc;
        B.fused_res = true;
        B.fused_k1 = k1;
      }
    }
    if (d.filters % 128)
      return fail(VASR_ERR_UNSUPPORTED, "block %zu: filters %d is not a multiple of 128", i, d.filters);
    cin = d.filters;
    // mid-pipeline buffers hold every block output but the last one, plus a last-block residual
    if ((i + 1 < h->blocks.size() || B.has_res) && cin > h->c_mid_max) h->c_mid_max = cin;
  }
  h->c_last = cin;
  return upload(h, h->steps, &h->d_steps);
}

int build_decoder(vasr_handle* h) {
  int rc;
  if ((rc = pack_pointwise(h, "decoder_layers.0.weight", h->num_classes, h->dec_feat_in, &h->dec))) return rc;
  const HostTensor* b;
  if ((rc = need(h, "decoder_layers.0.bias", h->num_classes, &b))) return rc;
  std::vector<float> sc(h->dec.m_pad, 1.f), sh(h->dec.m_pad, 0.f);
  for (int i = 0; i < h->num_classes; ++i) sh[i] = b->data[i];
  if ((rc = upload(h, sc, &h->dec.d_scale))) return rc;
  return upload(h, sh, &h->dec.d_shift);
}

// ---------------- workspace plan ----------------
struct WsPlan {
  size_t lens_tab, amax, seq, melp, bufP, bufQ, bufD, bufR, bufS, encp, logits, pred, total;
  int64_t T, Tp0, T1, Tp1;
  int amax_stride;   // slots per utterance of one maxima table (kAmaxTabs tables: [tab][B][amax_stride] u32)
};

int64_t enc_frames(const vasr_handle* h, int64_t t) {
  for (const Block& B : h->blocks)
    for (const SubBlock& S : B.subs)
      if (S.separable) t = conv_out_frames(t, S.dw);
  return t;
}

WsPlan plan_ws(const vasr_handle* h, int batch, int64_t T) {
  WsPlan p{};
  p.T = T;
  p.Tp0 = pad_frames(T);
  p.T1 = h->has_encoder ? enc_frames(h, T) : T;
  // the first (strided) block may still run at T frames inside: size by the larger pitch
  p.Tp1 = pad_frames(p.T1);
  const int64_t tp_mid = h->has_encoder ? pad_frames(conv_out_frames(T, h->blocks[0].subs[0].separable
                                                                          ? h->blocks[0].subs[0].dw
                                                                          : h->blocks[0].subs[0].pw))
                                        : p.Tp1;
  size_t o = 0;
  auto take = [&](size_t bytes) { size_t at = o; o = align_up(o + bytes, 256); return at; };
  p.lens_tab = take((h->steps.size() + 2) * (size_t)batch * 4);   // + the row-independent frame counts
  // maxima tables of the fp16-split arithmetic (AmaxTab): in flight at any time are the block input's, the current
  // kernel's input's and its output's -- kAmaxTabs = 4 rotate.  Capacity: the most slots any producer may use.
  p.amax_stride = 64;
  if (h->has_encoder) {
    int64_t t = T;
    for (const Block& B : h->blocks)
      for (const SubBlock& S : B.subs) {
        if (S.separable) {
          t = conv_out_frames(t, S.dw);
          p.amax_stride = std::max(p.amax_stride, depthwise_amax_slots(S.dw.cin, pad_frames(t)));
        }
        p.amax_stride = std::max(p.amax_stride, pointwise_amax_slots(S.pw.m_pad, pad_frames(t)));
      }
  }
  p.amax = take((size_t)kAmaxTabs * batch * p.amax_stride * 4);
  p.seq = take((size_t)batch * 8);
  p.melp = take((size_t)batch * (h->has_encoder ? h->feat_in : 64) * p.Tp0 * 4);
  const size_t mid = (size_t)batch * h->c_mid_max * std::max(tp_mid, p.Tp1) * 4;
  p.bufP = take(h->has_encoder ? mid : 0);
  p.bufQ = take(h->has_encoder ? mid : 0);
  p.bufD = take(h->has_encoder ? mid : 0);
  p.bufR = take(h->has_encoder ? mid : 0);
  p.bufS = take(h->has_encoder ? mid : 0);
  const int c_enc = h->has_encoder ? h->c_last : h->dec_feat_in;
  p.encp = take((size_t)batch * c_enc * p.Tp1 * 4);
  p.logits = take(h->has_decoder ? (size_t)batch * h->num_classes * p.Tp1 * 4 : 0);
  p.pred = take((size_t)batch * p.T1 * 8);
  p.total = o;
  return p;
}

// Brackets one launch (or a short launch group) with HIP events on the launch stream.
struct ProfScope {
  vasr_handle* h; hipStream_t st; hipEvent_t a{}, b{}; int cls; double flops, bytes;
  static hipEvent_t get(vasr_handle* h) {
    hipEvent_t e;
    if (!h->ev_pool.empty()) { e = h->ev_pool.back(); h->ev_pool.pop_back(); return e; }
    if (hipEventCreate(&e) != hipSuccess) e = nullptr;   // a null event makes the record / elapsed calls fail loudly
    return e;
  }
  // Single-launch classes (depthwise, pointwise) hand the event pair to the launch itself (g_probe: the dispatch
  // packet's begin / end timestamps); multi-launch groups (front end, head) are bracketed on the stream.
  // flops / bytes: the ALGORITHMIC work of the bracketed launch (2 M N K of a GEMM; read x + write y of a depthwise or
  // fused layer), summed per class by vasr_profile_end so that a rate is always work-that-ran over time-it-took
  ProfScope(vasr_handle* h_, int cls_, hipStream_t st_, double flops_ = 0.0, double bytes_ = 0.0)
      : h(h_), st(st_), cls(cls_), flops(flops_), bytes(bytes_) {
    if (!h->profiling) return;
    a = get(h); b = get(h);
    if (cls == 1 || cls == 2 || cls == 4) g_probe = LaunchProbe{a, b};
    else (void)hipEventRecord(a, st);
  }
  ~ProfScope() {
    if (!h->profiling) return;
    if (cls == 1 || cls == 2 || cls == 4) {
      if (g_probe.start) {   // no instrumented launch happened inside the scope
        g_probe = LaunchProbe{};
        (void)hipEventRecord(a, st);
        (void)hipEventRecord(b, st);
      }
    } else {
      (void)hipEventRecord(b, st);
    }
    h->prof.push_back({a, b, cls, flops, bytes});
  }
};
enum { kProfFrontend = 0, kProfDepthwise = 1, kProfPointwise = 2, kProfHead = 3, kProfFused = 4 };


int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(VASR_ERR_HIP, "%s launch: %s", what, hipGetErrorString(e));
  return 0;
}

// GEMM dispatch: exact-fp32 MFMA kernel, or the 3 x bf16 split kernel when selected and the layer has that pack.
// Returns 1 when the launch published a.amax_y (only the split kernel does), 0 when not, < 0 on error.
static int run_pointwise(vasr_handle* h, PwArgs& a, const ConvLayer& W, hipStream_t st) {
  if (h->gemm_mode >= 1 && W.d_w3) {
    int arith = h->gemm_mode == 2 ? 1 : 0;
    a.wt = reinterpret_cast<const float*>(W.d_w3);
    // fp16 split: needs the maxima of every source this GEMM reads; a source without them keeps the 3 x bf16 form
    if (h->gemm_mode == 3 && W.d_w16 && a.amax_x.p && (!a.x2 || a.amax_x2.p)) {
      arith = 2;
      a.wt = reinterpret_cast<const float*>(W.d_w16);
      a.w_inv_scale = W.w16_inv;
    }
    const int e = launch_pointwise_split(a, arith, st, &a.amax_y.n);
    if (e) return fail(VASR_ERR_HIP, "pointwise GEMM: %s", hipGetErrorString((hipError_t)e));
    return a.amax_y.p != nullptr ? 1 : 0;
  }
  a.wt = W.d_w;
  launch_pointwise(a, st);
  return 0;
}

// Encoder over an input [B][feat_in][x_ld]; writes [B][c_last][out_ld] (T1 valid frames).
// enc_amax (optional): receives the maxima table of the encoder output when one was published (fp16-split mode, padded
// output pitch), for the CTC head of the fused path; the table lives in the workspace until the next encoder pass.
int run_encoder(vasr_handle* h, const float* x, int64_t x_ld, int64_t T, const int64_t* seq, int batch,
                float* out, int64_t out_ld, float* enc_len, char* ws, const WsPlan& p, hipStream_t st,
                AmaxTab* enc_amax = nullptr, const int64_t* wav_len = nullptr, const int32_t** own_frames = nullptr,
                bool chain_done = false) {
  int32_t* lens_tab = reinterpret_cast<int32_t*>(ws + p.lens_tab);
  auto lens = [&](int step) { return lens_tab + (size_t)step * batch; };
  // row-independent mode (fused path): everything the CTC head sees of a row must depend on that row alone, also the
  // fp16 scale of its input -- the encoder output's maxima are then taken over the frames an unbatched call on the row
  // would produce, and the head zeroes the columns behind them (they are not decoded in this mode)
  const bool own = h->row_independent && wav_len != nullptr;
  // (the fused path has run the chain as extra workgroups of its normalisation launch: launch_normalize_chain)
  if (!chain_done)
    launch_len_chain(seq, batch, h->d_steps, (int)h->steps.size(), lens_tab, enc_len, st, own ? wav_len : nullptr,
                     h->fe.hop_length, (int)p.T1);
  const int32_t* own_tab = own ? lens((int)h->steps.size() + 1) : nullptr;
  if (own_frames) *own_frames = own_tab;
  float* bufs[4] = {reinterpret_cast<float*>(ws + p.bufP), reinterpret_cast<float*>(ws + p.bufQ),
                    reinterpret_cast<float*>(ws + p.bufR), reinterpret_cast<float*>(ws + p.bufS)};
  float* D = reinterpret_cast<float*>(ws + p.bufD);
  const float* cur = x;
  int64_t cur_ld = x_ld, cur_T = T;
  // fp16-split GEMMs: maxima rows, indexed by the length-chain step of the conv that produced the tensor
  const bool want_amax = h->gemm_mode == 3;
  unsigned int* amax_base = reinterpret_cast<unsigned int*>(ws + p.amax);
  AmaxTab cur_amax{}, blk_amax{};   // maxima of `cur` / of the block input over each utterance's valid frames (p == nullptr: not known)
  // a table that neither the block input's nor the current tensor's maxima live in (the third candidate is always free)
  auto free_tab = [&](const AmaxTab& also_busy) {
    for (int i = 0; i < kAmaxTabs; ++i) {
      unsigned int* q = amax_base + (size_t)i * batch * p.amax_stride;
      if (q != cur_amax.p && q != blk_amax.p && q != also_busy.p) return AmaxTab{q, p.amax_stride, 0};
    }
    return AmaxTab{};
  };
  for (size_t i = 0; i < h->blocks.size(); ++i) {
    Block& B = h->blocks[i];
    const bool last_block = i + 1 == h->blocks.size();
    const float* blk_in = cur;
    const int64_t blk_ld = cur_ld;
    blk_amax = cur_amax;
    // scratch buffers that are not the block input (the fused residual reads it until the block's last GEMM):
    // sub-block outputs ping-pong between the first two, an unfused residual result takes the third
    float* free3[3];
    int nf = 0;
    for (float* q : bufs) if (q != blk_in && nf < 3) free3[nf++] = q;
    float* R = free3[2];
    if (B.has_res && !B.fused_res) {
      // res branch: MaskedConv1d(1x1)(block input, lens_orig) -> BN   (parts/jasper.py:428-436)
      PwArgs a{};
      a.busy_cus = h->busy_cus;
      a.wt = B.res.d_w; a.x = cur; a.lens = lens(B.first_step); a.scale = B.res.d_scale; a.shift = B.res.d_shift;
      a.res = nullptr; a.y = R; a.M = B.res.m_pad; a.K = B.res.cin; a.batch = batch;
      a.ldx = cur_ld; a.ldy = cur_ld; a.ldr = 0; a.frames = (int)cur_T; a.store_cols = (int)cur_ld;
      a.m_store = B.res.m_pad; a.relu = 0;
      a.amax_x = blk_amax;
      ProfScope ps(h, kProfPointwise, st, 2.0 * B.res.cin * B.res.cout * (double)cur_T * batch,
                   4.0 * B.res.m_pad * (double)cur_ld * batch);   // bytes of class 2 = what the GEMM STORES (its epilogue's share of the time)
      if (run_pointwise(h, a, B.res, st) < 0) return VASR_ERR_HIP;
    }
    int flip = 0;
    for (size_t r = 0; r < B.subs.size(); ++r) {
      SubBlock& S = B.subs[r];
      const bool last_sub = r + 1 == B.subs.size();
      const float* gx = cur;
      int64_t gx_ld = cur_ld, g_T = cur_T;
      const int32_t* g_lens = nullptr;
      AmaxTab gx_amax = cur_amax;
      // ---- fused depthwise -> pointwise kernel (encoder_fused.hip): 256-channel sub-blocks in the fp16-split arithmetic, when
      //      there are enough 128-frame tiles to fill the chip (one workgroup per tile, all channels); VASR_FUSED=0 is the
      //      A/B switch ----
      const bool fused_on = h->sw.fused;
      // One workgroup per CU (159 KB of LDS), one 128-frame tile each: it pays when the tiles fill whole rounds of the chip
      // (measured, fused vs two kernels: 64 x 10 s = 256 tiles -3.4 % per step, 512 x 30 s = 6144 tiles -4 %; but 32 x 10 s = 128
      // tiles +2.6 %, 64 x 10.3 s = 320 tiles = 1.25 rounds +3 %, 16 x 10 s = 64 tiles +4.6 %).  Smaller batches take the kernel's
      // 64-frame form (round 4) while THOSE tiles fit one round and occupy at least 3/8 of the chip -- 12 to 32 utterances of
      // 10 s; measured against two kernels, ms per step: 15x5 B = 12 / 16 / 20 / 24 / 32: -3.8 / -4.8 / -8.3 / -7.7 / -6.7 %, 12x1
      // (BASELINE configs[1]) B = 16 / 20 / 24 / 32: -3.1 / -5.1 / -5.0 / -4.3 %; B = 8: +-0; B = 1-4: +5 ... +9 % (one tile is 14 us
      // of latency against 4 + 7 us for the two kernels spread over the chip); 36-44 utterances (1.1-1.4 rounds): +0.8 ... +1.6 %.
      const int fused_min_tiles = h->sw.fused_min_tiles, fused_tile = h->sw.fused_tile;
      const int n_cu = h->n_cu;
      // (units a concurrent kernel of the caller's holds -- the overlapped beam search -- take no workgroups: 256 tiles on the
      // 192 CUs a 64-utterance search leaves free are 1.33 rounds)
      const int f_cus = h->busy_cus > 0 && h->busy_cus < n_cu - 32 ? n_cu - h->busy_cus : n_cu;
      const int64_t f_tiles = (int64_t)batch * (cur_ld / kTimeTile);
      // WHETHER a sub-block is fused is a function of the batch's shape ALONE (the whole chip's unit count): the fused and the
      // two-kernel form round differently, and the busy-unit hint follows a concurrent kernel's progress -- with the hint in this
      // decision the log-probs of one and the same batch depended on whether the previous search had finished (round 6,
      // tests/devtools/stress_beam_overlap.py: 196 of 11 594 overlapped batches differed from their serial run).  The hint only
      // picks the TILE WIDTH of a sub-block that is fused anyway: 64- and 128-frame tiles give the same bits.
      const int f_rule = fused_tile_choice(f_tiles, n_cu);
      int f_auto = f_rule ? f_rule : 128;
      if (f_rule && f_cus < n_cu) {
        // lock-stepped rounds on the free units: a 64-frame tile costs ~0.6 of a 128-frame one (17 vs 29.7 us a round); 256 tiles on
        // the 192 units a 64-utterance search leaves: 2 rounds of 128 frames = 2.0 against 3 rounds of 64 = 1.8
        const int64_t r128 = (f_tiles + f_cus - 1) / f_cus, r64 = (2 * f_tiles + f_cus - 1) / f_cus;
        f_auto = 0.6 * (double)r64 < (double)r128 ? 64 : 128;
      }
      const int f_cols = fused_tile == 64 || fused_tile == 128 ? fused_tile : f_auto;
      const bool f_fill = fused_min_tiles > 0 ? f_tiles >= fused_min_tiles : (fused_tile ? f_rule == fused_tile : f_rule != 0);
      const bool fuse_res = last_sub && B.fused_res;
      const ConvLayer& WF = fuse_res ? B.fused : S.pw;
      // (not in row-independent mode: whether a sub-block is fused depends on the batch's tile count, and the two forms
      // round differently -- that mode promises bit-identical rows whatever the batch)
      if (fused_on && !h->row_independent && S.separable && S.dw.d_ftaps && h->gemm_mode == 3 && want_amax && cur_amax.p && WF.d_w16 &&
          !(last_sub && B.has_res && !B.fused_res) &&
          // a folded residual must come from a 256-channel block input (K = 256 + 256): the kernel's second K range is 4 chunks
          (fuse_res ? (blk_amax.p && blk_ld == cur_ld && WF.cin == 2 * S.dw.cin && B.fused_k1 == S.dw.cin) : WF.cin == S.dw.cin) &&
          cur_ld % kTimeTile == 0 && f_fill &&
          !(last_block && last_sub)) {
        float* dst = free3[flip];
        flip ^= 1;
        FusedLaunch f{};
        f.x = cur; f.ldx = cur_ld; f.lens_in = lens(S.dw.step); f.lens_out = lens(S.dw.step + 1);
        f.taps = S.dw.d_ftaps; f.dw_l1 = S.dw.f_l1; f.amax_x = cur_amax;
        f.wt = WF.d_w16; f.w_inv_scale = WF.w16_inv; f.scale = WF.d_scale; f.shift = WF.d_shift;
        f.y = dst; f.ldy = cur_ld; f.frames = (int)cur_T; f.relu = 1;
        f.amax_y = free_tab(cur_amax); f.lens_y = lens(S.pw.step + 1);
        if (fuse_res) { f.x2 = blk_in; f.ldx2 = blk_ld; f.lens2 = lens(B.first_step); f.amax_x2 = blk_amax; }
        f.batch = batch; f.kernel = S.dw.kernel;
        f.tile_cols = f_cols;
        int e;
        {
          ProfScope ps(h, kProfFused, st, 2.0 * WF.cin * WF.cout * (double)cur_T * batch,
                       4.0 * (2.0 + (fuse_res ? 1.0 : 0.0)) * S.dw.cin * (double)cur_T * batch);
          e = launch_fused_dwpw(f, st, &f.amax_y.n);
        }
        if (e > 0) return fail(VASR_ERR_HIP, "fused dw -> pw: %s", hipGetErrorString((hipError_t)e));
        if (e == 0) {
          cur = dst;
          cur_amax = f.amax_y;
          continue;
        }
        flip ^= 1;   // shape not covered after all: the two-kernel path below
      }
      if (S.separable) {
        const int64_t t_out = conv_out_frames(cur_T, S.dw);
        const int64_t ld_out = pad_frames(t_out);
        ProfScope ps(h, kProfDepthwise, st, 2.0 * S.dw.kernel * S.dw.cin * (double)t_out * batch,
                     4.0 * ((double)S.dw.cin * cur_T + (double)S.dw.cin * t_out) * batch + 4.0 * S.dw.cin * S.dw.kernel);
        AmaxTab am = want_amax ? free_tab(AmaxTab{}) : AmaxTab{};
        // fp16-split mode with the input's maxima at hand: the Toeplitz form on the matrix pipe (in the pipeline, per
        // 512-channel layer: 22.8 / 23.7 / 23.8 / 28.1 us at K = 51 / 63 / 75 / 87 x 2 against 25.6 / 28.4 / 31.0 / 35.5 us
        // of packed FMAs; 256 channels, K = 33 / 39: 12.8 / 12.6 against 13.3 / 13.4), else packed FMAs.
        // VASR_DW_MFMA=0 keeps the packed-FMA kernels.
        const bool dw_mfma = h->sw.dw_mfma;
        int e = -1;
        if (want_amax && dw_mfma && cur_amax.p && S.dw.d_taps)
          e = launch_depthwise_mfma(cur, cur_ld, S.dw.d_taps, S.dw.d_tap_inv, lens(S.dw.step), lens(S.dw.step + 1), cur_amax,
                                    batch, S.dw.cin, S.dw.kernel, S.dw.dilation, D, ld_out, am.p ? &am : nullptr, st);
        if (e > 0) return fail(VASR_ERR_HIP, "depthwise (MFMA): %s", hipGetErrorString((hipError_t)e));
        if (e < 0 && launch_depthwise(cur, cur_ld, (int)cur_T, S.dw.d_w, lens(S.dw.step), lens(S.dw.step + 1), batch, S.dw.cin,
                                      S.dw.kernel, S.dw.stride, S.dw.dilation, S.dw.pad, D, ld_out, st, am.p ? &am : nullptr))
          return fail(VASR_ERR_WORKSPACE, "maxima table too small for depthwise layer of block %zu", i);
        gx = D; gx_ld = ld_out; g_T = t_out; gx_amax = am;
      } else {
        g_lens = lens(S.pw.step);  // block input is unmasked: predicate inside the GEMM
      }
      float* dst = free3[flip];
      flip ^= 1;
      int64_t dst_ld = gx_ld;
      if (last_block && last_sub) { dst = out; dst_ld = out_ld; }
      const bool fuse = last_sub && B.fused_res;
      const ConvLayer& W = fuse ? B.fused : S.pw;
      PwArgs a{};
      a.busy_cus = h->busy_cus;
      a.wt = W.d_w; a.x = gx; a.lens = g_lens; a.scale = W.d_scale; a.shift = W.d_shift;
      a.res = (last_sub && B.has_res && !B.fused_res) ? R : nullptr;
      a.y = dst; a.M = W.m_pad; a.K = W.cin; a.batch = batch;
      a.ldx = gx_ld; a.ldy = dst_ld; a.ldr = blk_ld; a.frames = (int)g_T; a.m_store = W.m_pad; a.relu = 1;
      a.store_cols = (dst_ld % kTimeTile == 0) ? (int)dst_ld : (int)g_T;  // port tensors are not padded
      // the depthwise output is zero past its lens_out, a masked input past its mask: tiles out there skip their K loop
      a.zero_from = S.separable ? lens(S.dw.step + 1) : g_lens;
      if (fuse) { a.x2 = blk_in; a.lens2 = lens(B.first_step); a.K1 = B.fused_k1; a.ldx2 = blk_ld; }
      if ((a.res || fuse) && blk_ld != gx_ld)
        return fail(VASR_ERR_UNSUPPORTED, "block %zu: residual across a strided block", i);
      a.amax_x = gx_amax;
      a.amax_x2 = fuse ? blk_amax : AmaxTab{};
      // this GEMM's output is masked at lens(S.pw.step + 1) by whatever reads it next
      // (the encoder output's maxima are only wanted by the fused path's CTC head, which reads every column < T')
      if (want_amax && (!(last_block && last_sub) || enc_amax)) {
        a.amax_y = free_tab(gx_amax);
        a.lens_y = (last_block && last_sub) ? own_tab : lens(S.pw.step + 1);
      }
      int published;
      {
        ProfScope ps(h, kProfPointwise, st, 2.0 * W.cin * W.cout * (double)g_T * batch, 4.0 * W.m_pad * (double)dst_ld * batch);
        published = run_pointwise(h, a, W, st);
      }
      if (published < 0) return VASR_ERR_HIP;
      cur = dst; cur_ld = dst_ld; cur_T = g_T;
      cur_amax = published ? a.amax_y : AmaxTab{};
    }
  }
  if (enc_amax) *enc_amax = cur_amax;
  return check_launch("encoder");
}

int run_decoder(vasr_handle* h, const float* encp, int64_t ld, int64_t T1, int batch, float* logits, float* logp,
                int64_t* pred, hipStream_t st, AmaxTab enc_amax = AmaxTab{}, const int32_t* own_frames = nullptr) {
  PwArgs a{};
  a.busy_cus = h->busy_cus;
  // own_frames (row-independent mode): columns past the row's own frame count are read as zero
  a.wt = h->dec.d_w; a.x = encp; a.lens = own_frames; a.scale = h->dec.d_scale; a.shift = h->dec.d_shift;
  a.res = nullptr; a.y = logits; a.M = h->dec.m_pad; a.K = h->dec.cin; a.batch = batch;
  a.ldx = ld; a.ldy = ld; a.ldr = 0; a.frames = (int)T1; a.store_cols = (int)ld; a.m_store = h->num_classes;
  a.relu = 0;
  ProfScope ps(h, kProfHead, st);
  // split GEMM unless fp32 mode is selected: the fp16 form when the encoder published its output's maxima (fused path),
  // else 3 x bf16 (port tensors of the per-module entry point carry none)
  a.amax_x = enc_amax;
  if (run_pointwise(h, a, h->dec, st) < 0) return VASR_ERR_HIP;
  launch_logsoftmax_argmax(logits, ld, (int64_t)h->num_classes * ld, batch, (int)T1, h->num_classes, logp, pred, st);
  return check_launch("decoder");
}

}  // namespace

extern "C" {

const char* vasr_last_error(void) { return g_err.c_str(); }
const char* vasr_version(void) { return "vasr-hip 0.4 (gfx950)"; }
int vasr_abi_version(void) { return VASR_ABI_VERSION; }
int64_t vasr_padded_frames(int64_t frames) { return pad_frames(frames); }
#ifdef VASR_DEVTOOLS
int vasr_fused_tile_choice(int64_t tiles128, int compute_units) { return vasr::fused_tile_choice(tiles128, compute_units); }
#endif

int vasr_create(const vasr_model_desc* d, vasr_handle** out) {
  if (!d || !out) return fail(VASR_ERR_INVALID, "null argument");
  auto* h = new vasr_handle();
  if (d->frontend) {
    const vasr_frontend_desc& fe = *d->frontend;
    if (fe.n_fft != 512 || fe.n_mels != 64) {
      delete h;
      return fail(VASR_ERR_UNSUPPORTED, "front end supports n_fft=512, n_mels=64 (got %d, %d)", fe.n_fft, fe.n_mels);
    }
    if (fe.win_length <= 0 || fe.win_length > fe.n_fft || fe.hop_length <= 0 || fe.hop_length > 512 ||
        !fe.h_filterbank) {
      delete h;
      // FilterbankFeatures.__init__ raises ValueError for non-positive window sizes (features.py:137-149)
      return fail(VASR_ERR_INVALID, "invalid window/hop/filterbank in front-end description");
    }
    h->has_frontend = true;
    h->fe = fe;
    h->fe_window.resize(fe.win_length);
    for (int i = 0; i < fe.win_length; ++i)
      h->fe_window[i] = fe.h_window ? fe.h_window[i]
                                    : (float)(0.5 - 0.5 * std::cos(2.0 * M_PI * i / (fe.win_length - 1)));
    h->fe_fb.assign(fe.h_filterbank, fe.h_filterbank + (size_t)fe.n_mels * (fe.n_fft / 2 + 1));
    h->fe.h_window = nullptr;
    h->fe.h_filterbank = nullptr;
  }
  if (d->n_blocks > 0) {
    if (!d->blocks || d->feat_in <= 0) { delete h; return fail(VASR_ERR_INVALID, "encoder needs blocks and feat_in"); }
    h->has_encoder = true;
    h->feat_in = d->feat_in;
    for (int i = 0; i < d->n_blocks; ++i) {
      const vasr_block_desc& b = d->blocks[i];
      if (b.filters <= 0 || b.repeat <= 0 || b.kernel <= 0 || b.stride <= 0 || b.dilation <= 0) {
        delete h;
        return fail(VASR_ERR_INVALID, "block %d has a non-positive field", i);
      }
      Block B;
      B.d = b;
      h->blocks.push_back(B);
    }
  }
  if (d->num_classes > 0) {
    if (d->dec_feat_in <= 0 || d->num_classes > 128) {
      delete h;
      return fail(VASR_ERR_UNSUPPORTED, "decoder needs dec_feat_in > 0 and at most 128 classes incl. blank");
    }
    h->has_decoder = true;
    h->dec_feat_in = d->dec_feat_in;
    h->num_classes = d->num_classes;
  }
  *out = h;
  return 0;
}

void vasr_destroy(vasr_handle* h) {
  if (!h) return;
  for (void* p : h->dev_allocs) (void)hipFree(p);
  for (auto& r : h->prof) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
  for (hipEvent_t e : h->ev_pool) (void)hipEventDestroy(e);
  if (h->slice_ready) {
    for (int i = 0; i < kMaxSlices; ++i) { (void)hipStreamDestroy(h->slice_stream[i]); (void)hipEventDestroy(h->slice_done[i]); }
    (void)hipEventDestroy(h->slice_fork);
  }
  delete h;
}

int vasr_load_weight(vasr_handle* h, const char* key, const float* data, const int64_t* shape, int ndim) {
  if (!h || !key || (!data && ndim > 0)) return fail(VASR_ERR_INVALID, "null argument");
  if (h->finalized) return fail(VASR_ERR_STATE, "handle already finalized");
  const size_t n = strlen(key);
  if (n >= 19 && strcmp(key + n - 19, "num_batches_tracked") == 0) return 0;
  size_t numel = 1;
  HostTensor t;
  for (int i = 0; i < ndim; ++i) {
    if (shape[i] < 0) return fail(VASR_ERR_INVALID, "negative dimension in '%s'", key);
    numel *= (size_t)shape[i];
    t.shape.push_back(shape[i]);
  }
  t.data.assign(data, data + numel);
  h->weights[key] = std::move(t);
  return 0;
}

int vasr_finalize(vasr_handle* h) {
  if (!h) return fail(VASR_ERR_INVALID, "null handle");
  if (h->finalized) return 0;
  int rc;
  if (h->has_frontend && (rc = build_frontend(h))) return rc;
  if (h->has_encoder && (rc = build_encoder(h))) return rc;
  if (h->has_decoder && (rc = build_decoder(h))) return rc;
  HIP_TRY(hipDeviceSynchronize());
  {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0)
      h->n_cu = n;
  }
  h->weights.clear();
  h->finalized = true;
  return 0;
}

int64_t vasr_mel_frames(const vasr_handle* h, int64_t samples) {
  const int hop = (h && h->has_frontend) ? h->fe.hop_length : 160;
  return 1 + samples / hop;
}

int64_t vasr_encoded_frames(const vasr_handle* h, int64_t mel_frames) {
  if (!h || !h->has_encoder || !h->finalized) return mel_frames;
  return enc_frames(h, mel_frames);
}

static size_t sliced_workspace_bytes(const vasr_handle* h, int batch, int64_t T);

size_t vasr_workspace_bytes(const vasr_handle* h, int batch, int64_t samples, int64_t mel_frames) {
  if (!h || !h->finalized || batch <= 0) return 0;
  const int64_t T = samples > 0 ? vasr_mel_frames(h, samples) : mel_frames;
  const size_t whole = plan_ws(h, batch, T).total;
  if (samples > 0 && h->has_frontend && h->has_encoder && h->has_decoder)
    return std::max(whole, sliced_workspace_bytes(h, batch, T));
  return whole;
}

int vasr_melspec_f32(vasr_handle* h, const float* d_wav, const int64_t* d_len, int batch, int64_t samples,
                     float* d_mel, int64_t* d_seq, vasr_stream stream) {
  if (!h || !h->has_frontend || !h->finalized) return fail(VASR_ERR_STATE, "no finalized front end in this handle");
  if (batch <= 0 || !d_wav || !d_len || !d_mel || !d_seq) return fail(VASR_ERR_INVALID, "bad argument");
  if (samples <= h->fe.n_fft / 2)
    return fail(VASR_ERR_INVALID, "reflect padding needs more than n_fft/2 = %d samples (got %lld)",
                h->fe.n_fft / 2, (long long)samples);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int T = (int)vasr_mel_frames(h, samples);
  launch_seq_len(d_len, batch, h->fe.hop_length, d_seq, st);
  launch_stft_logmel(h->ft, d_wav, false, batch, samples, h->row_independent ? d_len : nullptr, h->fe.hop_length,
                     h->fe.preemph, h->fe.log_guard, d_mel, T, T, st);
  launch_normalize(d_mel, T, d_seq, batch, h->fe.n_mels, T, h->fe.normalize == 1, st);
  if (h->fe.normalize == 2) launch_normalize_all(d_mel, T, d_seq, batch, h->fe.n_mels, T, st);
  return check_launch("melspec");
}

int vasr_encoder_f32(vasr_handle* h, const float* d_mel, const int64_t* d_seq, int batch, int64_t mel_frames,
                     float* d_enc, float* d_enc_len, void* d_ws, size_t ws_bytes, vasr_stream stream) {
  if (!h || !h->has_encoder || !h->finalized) return fail(VASR_ERR_STATE, "no finalized encoder in this handle");
  if (batch <= 0 || mel_frames <= 0 || !d_mel || !d_seq || !d_enc || !d_ws) return fail(VASR_ERR_INVALID, "bad argument");
  const WsPlan p = plan_ws(h, batch, mel_frames);
  if (ws_bytes < p.total) return fail(VASR_ERR_WORKSPACE, "workspace %zu < %zu bytes", ws_bytes, p.total);
  hipStream_t st = static_cast<hipStream_t>(stream);
  char* ws = static_cast<char*>(d_ws);
  const Block& b0 = h->blocks[0];
  if (b0.subs[0].separable && !b0.has_res)
    // block 0 reads the port tensor in place (the generic depthwise kernel copes with the unpadded pitch)
    return run_encoder(h, d_mel, mel_frames, mel_frames, d_seq, batch, d_enc, p.T1, d_enc_len, ws, p, st);
  float* melp = reinterpret_cast<float*>(ws + p.melp);
  launch_repad(d_mel, mel_frames, batch * h->feat_in, (int)mel_frames, melp, p.Tp0, st);
  return run_encoder(h, melp, p.Tp0, mel_frames, d_seq, batch, d_enc, p.T1, d_enc_len, ws, p, st);
}

int vasr_decoder_logsoftmax_f32(vasr_handle* h, const float* d_enc, int batch, int64_t enc_frames_, float* d_logp,
                                void* d_ws, size_t ws_bytes, vasr_stream stream) {
  if (!h || !h->has_decoder || !h->finalized) return fail(VASR_ERR_STATE, "no finalized decoder in this handle");
  if (batch <= 0 || enc_frames_ <= 0 || !d_enc || !d_logp || !d_ws) return fail(VASR_ERR_INVALID, "bad argument");
  const int64_t ld = pad_frames(enc_frames_);
  const size_t enc_bytes = align_up((size_t)batch * h->dec_feat_in * ld * 4, 256);
  const size_t need_bytes = enc_bytes + (size_t)batch * h->num_classes * ld * 4;
  if (ws_bytes < need_bytes) return fail(VASR_ERR_WORKSPACE, "workspace %zu < %zu bytes", ws_bytes, need_bytes);
  hipStream_t st = static_cast<hipStream_t>(stream);
  float* encp = static_cast<float*>(d_ws);
  float* logits = reinterpret_cast<float*>(static_cast<char*>(d_ws) + enc_bytes);
  launch_repad(d_enc, enc_frames_, batch * h->dec_feat_in, (int)enc_frames_, encp, ld, st);
  // The fused path's head runs in the fp16-split arithmetic on the maxima the last encoder GEMM published; a port tensor carries
  // none, and rounds 2-5 ran this entry point as 3 x bf16 instead -- same tolerance, other bits: of 181 272 signals 38 came out of
  // the reference-style module-by-module path one character different from the fused row-independent path that promises "what the
  // signal alone gives" (round 6, tests/devtools/fuzz_dag.py).  Now the maxima are taken here, over the same frames, when the
  // caller's workspace has the 1 KB per utterance for them (vasr.h): the two paths give the same bits.
  AmaxTab ax{};
  const size_t amax_off = align_up(need_bytes, 256), amax_bytes = (size_t)batch * 256 * 4;
  if (h->gemm_mode == 3 && ws_bytes >= amax_off + amax_bytes) {
    ax = AmaxTab{reinterpret_cast<unsigned int*>(static_cast<char*>(d_ws) + amax_off), 256, 0};
    launch_amax(encp, ld, h->dec_feat_in, (int)enc_frames_, nullptr, batch, &ax, st);
  }
  return run_decoder(h, encp, ld, enc_frames_, batch, logits, d_logp, nullptr, st, ax);
}

int vasr_greedy_argmax(const float* d_logp, int batch, int64_t frames, int num_classes, int64_t* d_pred,
                       vasr_stream stream) {
  if (!d_logp || !d_pred || batch <= 0 || frames <= 0 || num_classes <= 0) return fail(VASR_ERR_INVALID, "bad argument");
  launch_argmax(d_logp, batch, frames, num_classes, d_pred, static_cast<hipStream_t>(stream));
  return check_launch("argmax");
}

int vasr_ctc_collapse(const int64_t* d_pred, int batch, int64_t frames, int blank_id, int32_t* d_ids,
                      int32_t* d_id_len, vasr_stream stream) {
  if (!d_pred || !d_ids || !d_id_len || batch <= 0 || frames < 0) return fail(VASR_ERR_INVALID, "bad argument");
  launch_ctc_collapse(d_pred, batch, frames, blank_id, d_ids, d_id_len, static_cast<hipStream_t>(stream));
  return check_launch("ctc_collapse");
}

// One contiguous slice of the batch through the whole path on one stream.
static int transcribe_part(vasr_handle* h, const void* d_wav, bool pcm16, const int64_t* d_len, int batch, int64_t samples,
                           int64_t* d_pred, int32_t* d_ids, int32_t* d_id_len, float* d_logp, float* d_enc_len,
                           char* ws, hipStream_t st) {
  const int64_t T = vasr_mel_frames(h, samples);
  const WsPlan p = plan_ws(h, batch, T);
  int64_t* seq = reinterpret_cast<int64_t*>(ws + p.seq);
  float* melp = reinterpret_cast<float*>(ws + p.melp);
  float* encp = reinterpret_cast<float*>(ws + p.encp);
  float* logits = reinterpret_cast<float*>(ws + p.logits);
  int64_t* pred = d_pred ? d_pred : reinterpret_cast<int64_t*>(ws + p.pred);
  {
    // two launches: the STFT / log-mel tiles, then the per-feature normalisation whose launch also carries (as extra
    // workgroups) seq = ceil(len / hop) and the encoder's length chain -- three launches fewer on the critical path of a
    // batch-1 call than seq_len + stft + normalize + len_chain
    ProfScope ps(h, kProfFrontend, st);
    launch_stft_logmel(h->ft, d_wav, pcm16, batch, samples, h->row_independent ? d_len : nullptr, h->fe.hop_length,
                       h->fe.preemph, h->fe.log_guard, melp, p.Tp0, (int)T, st);
    launch_normalize_chain(melp, p.Tp0, d_len, h->fe.hop_length, batch, h->fe.n_mels, (int)T, h->fe.normalize == 1, seq,
                           h->d_steps, (int)h->steps.size(), reinterpret_cast<int32_t*>(ws + p.lens_tab), d_enc_len,
                           h->row_independent ? d_len : nullptr, (int)p.T1, st);
    if (h->fe.normalize == 2) launch_normalize_all(melp, p.Tp0, seq, batch, h->fe.n_mels, (int)T, st);
  }
  AmaxTab enc_amax{};
  const int32_t* own_frames = nullptr;
  int rc = run_encoder(h, melp, p.Tp0, T, seq, batch, encp, p.Tp1, d_enc_len, ws, p, st, &enc_amax, d_len, &own_frames, true);
  if (rc) return rc;
  if ((rc = run_decoder(h, encp, p.Tp1, p.T1, batch, logits, d_logp, pred, st, enc_amax, own_frames))) return rc;
  if (d_ids && d_id_len) {
    ProfScope ps(h, kProfHead, st);
    if (h->row_independent)
      launch_ctc_collapse(pred, batch, p.T1, h->num_classes - 1, d_ids, d_id_len, st, d_len, h->fe.hop_length,
                          h->d_steps, (int)h->steps.size());
    else
      launch_ctc_collapse(pred, batch, p.T1, h->num_classes - 1, d_ids, d_id_len, st);
  }
  return 0;
}

// How many slices the batch is cut into; slice i runs on its own internal stream so that one slice's
// HBM-bound kernels (depthwise, epilogue stores) overlap another slice's MFMA-bound GEMM main loops.
static int n_slices(const vasr_handle* h, int batch) {
  int n = h->slices;
  if (n < 1) n = 1;
  if (n > kMaxSlices) n = kMaxSlices;
  while (n > 1 && batch / n < 8) --n;   // tiny batches: one slice (the kernels would not fill the chip anyway)
  return n;
}

static size_t sliced_workspace_bytes(const vasr_handle* h, int batch, int64_t T) {
  const int n = n_slices(h, batch);
  size_t total = 0;
  for (int i = 0; i < n; ++i) {
    const int lo = (int)((int64_t)batch * i / n), hi = (int)((int64_t)batch * (i + 1) / n);
    total += align_up(plan_ws(h, hi - lo, T).total, 256);
  }
  return total;
}

static int transcribe_any(vasr_handle* h, const void* d_wav, bool pcm16, const int64_t* d_len, int batch, int64_t samples,
                          int64_t* d_pred, int32_t* d_ids, int32_t* d_id_len, float* d_logp, float* d_enc_len,
                          void* d_ws, size_t ws_bytes, vasr_stream stream) {
  if (!h || !h->finalized || !h->has_frontend || !h->has_encoder || !h->has_decoder)
    return fail(VASR_ERR_STATE, "handle needs a finalized front end, encoder and decoder");
  if (batch <= 0 || !d_wav || !d_len || !d_ws) return fail(VASR_ERR_INVALID, "bad argument");
  if (samples <= h->fe.n_fft / 2)
    return fail(VASR_ERR_INVALID, "reflect padding needs more than n_fft/2 = %d samples (got %lld)",
                h->fe.n_fft / 2, (long long)samples);
  const int64_t T = vasr_mel_frames(h, samples);
  const size_t need_bytes = sliced_workspace_bytes(h, batch, T);
  if (ws_bytes < need_bytes) return fail(VASR_ERR_WORKSPACE, "workspace %zu < %zu bytes", ws_bytes, need_bytes);
  hipStream_t user = static_cast<hipStream_t>(stream);
  char* ws = static_cast<char*>(d_ws);
  const int n = n_slices(h, batch);
  const int64_t T1 = vasr_encoded_frames(h, T);
  if (n == 1) {
    int rc = transcribe_part(h, d_wav, pcm16, d_len, batch, samples, d_pred, d_ids, d_id_len, d_logp, d_enc_len, ws, user);
    return rc ? rc : check_launch("transcribe");
  }
  if (!h->slice_ready) {
    for (int i = 0; i < kMaxSlices; ++i) {
      HIP_TRY(hipStreamCreateWithFlags(&h->slice_stream[i], hipStreamNonBlocking));
      HIP_TRY(hipEventCreateWithFlags(&h->slice_done[i], hipEventDisableTiming));
    }
    HIP_TRY(hipEventCreateWithFlags(&h->slice_fork, hipEventDisableTiming));
    h->slice_ready = true;
  }
  HIP_TRY(hipEventRecord(h->slice_fork, user));
  size_t off = 0;
  for (int i = 0; i < n; ++i) {
    const int lo = (int)((int64_t)batch * i / n), hi = (int)((int64_t)batch * (i + 1) / n);
    hipStream_t st = h->slice_stream[i];
    HIP_TRY(hipStreamWaitEvent(st, h->slice_fork, 0));
    int rc = transcribe_part(h, static_cast<const char*>(d_wav) + (int64_t)lo * samples * (pcm16 ? 2 : 4), pcm16, d_len + lo, hi - lo, samples,
                             d_pred ? d_pred + (int64_t)lo * T1 : nullptr, d_ids ? d_ids + (int64_t)lo * T1 : nullptr,
                             d_id_len ? d_id_len + lo : nullptr,
                             d_logp ? d_logp + (int64_t)lo * T1 * h->num_classes : nullptr,
                             d_enc_len ? d_enc_len + lo : nullptr, ws + off, st);
    if (rc) return rc;
    off += align_up(plan_ws(h, hi - lo, T).total, 256);
    HIP_TRY(hipEventRecord(h->slice_done[i], st));
    HIP_TRY(hipStreamWaitEvent(user, h->slice_done[i], 0));
  }
  return check_launch("transcribe");
}

int vasr_transcribe_greedy_f32(vasr_handle* h, const float* d_wav, const int64_t* d_len, int batch, int64_t samples,
                               int64_t* d_pred, int32_t* d_ids, int32_t* d_id_len, float* d_logp, float* d_enc_len,
                               void* d_ws, size_t ws_bytes, vasr_stream stream) {
  return transcribe_any(h, d_wav, false, d_len, batch, samples, d_pred, d_ids, d_id_len, d_logp, d_enc_len, d_ws, ws_bytes, stream);
}

int vasr_transcribe_greedy_pcm16(vasr_handle* h, const int16_t* d_pcm, const int64_t* d_len, int batch, int64_t samples,
                                 int64_t* d_pred, int32_t* d_ids, int32_t* d_id_len, float* d_logp, float* d_enc_len,
                                 void* d_ws, size_t ws_bytes, vasr_stream stream) {
  return transcribe_any(h, d_pcm, true, d_len, batch, samples, d_pred, d_ids, d_id_len, d_logp, d_enc_len, d_ws, ws_bytes, stream);
}

int vasr_pcm16_to_f32(const int16_t* d_pcm, int64_t n, float* d_out, vasr_stream stream) {
  if (!d_pcm || !d_out || n < 0) return fail(VASR_ERR_INVALID, "bad argument");
  if (n == 0) return 0;
  if ((reinterpret_cast<uintptr_t>(d_pcm) & 7) || (reinterpret_cast<uintptr_t>(d_out) & 15))
    return fail(VASR_ERR_INVALID, "pcm / float buffers must be 8 / 16-byte aligned");
  launch_pcm16_to_f32(d_pcm, n, d_out, static_cast<hipStream_t>(stream));
  return check_launch("pcm16_to_f32");
}

int vasr_resample_f32(const float* d_in, int64_t ld_in, const int64_t* d_len_in, int batch, const float* d_table,
                      int nwin, int num_table, double ratio, float* d_out, int64_t ld_out, int64_t* d_len_out,
                      vasr_stream stream) {
  if (!d_in || !d_len_in || !d_table || !d_out || !d_len_out || batch <= 0 || ld_in <= 0 || ld_out <= 0)
    return fail(VASR_ERR_INVALID, "bad argument");
  if (!(ratio > 0.0) || nwin < 2 || num_table < 1 || (int)((ratio < 1.0 ? ratio : 1.0) * num_table) < 1)
    return fail(VASR_ERR_INVALID, "invalid ratio / table for resampling");
  launch_resample(d_in, ld_in, d_len_in, batch, d_table, nwin, num_table, ratio, d_out, ld_out, d_len_out,
                  static_cast<hipStream_t>(stream));
  return check_launch("resample");
}

int vasr_set_gemm_mode(vasr_handle* h, int mode) {
  if (!h || mode < 0 || mode > 3)
    return fail(VASR_ERR_INVALID, "gemm mode must be 0 (fp32 MFMA), 1 (3 x bf16 split), 2 (2 x bf16 split, reduced) or "
                                  "3 (2 x fp16 scaled split)");
  h->gemm_mode = mode;
  return 0;
}

int vasr_get_gemm_mode(const vasr_handle* h) { return h ? h->gemm_mode : -1; }

int vasr_set_busy_cus(vasr_handle* h, int cus) {
  if (!h || cus < 0) return fail(VASR_ERR_INVALID, "busy compute units must be >= 0");
  h->busy_cus = cus;
  return 0;
}

int vasr_set_row_independent(vasr_handle* h, int on) {
  if (!h) return fail(VASR_ERR_INVALID, "null handle");
  h->row_independent = on != 0;
  return 0;
}

int vasr_set_slices(vasr_handle* h, int slices) {
  if (!h || slices < 1 || slices > kMaxSlices) return fail(VASR_ERR_INVALID, "slices must be 1..%d", kMaxSlices);
  h->slices = slices;
  return 0;
}

size_t vasr_beam_workspace_bytes(int batch, int64_t frames) {
  // back-pointer rows [B][T][128] u32, then the LM-cache key log [B][T * 128] u64 (beam_wave.hip: eoslog)
  return batch > 0 && frames > 0 ? (size_t)batch * frames * kBeamMax * (sizeof(unsigned int) + sizeof(uint64_t)) : 0;
}

int vasr_beam_workgroups(int batch) {
  if (batch <= 0) return 0;
  if (beam_group_width(batch) > 1) return batch;     // the latency form: a compute unit per utterance (beam_group.hip)
  const int upw = beam_wave_utts_per_workgroup(batch);
  return (batch + upw - 1) / upw;
}

int vasr_beam_search_f32(const float* d_logp, int batch, int64_t frames, int num_classes, int space_id,
                         int beam_width, float token_min_logp, float beam_prune_logp, const vasr_lm* lm,
                         int32_t* d_ids, int32_t* d_id_len, float* d_score, void* d_ws, size_t ws_bytes,
                         vasr_stream stream) {
  return vasr_beam_search_rows_f32(d_logp, nullptr, batch, frames, num_classes, space_id, beam_width, token_min_logp,
                                   beam_prune_logp, lm, d_ids, d_id_len, d_score, d_ws, ws_bytes, stream);
}

int vasr_beam_search_rows_f32(const float* d_logp, const int32_t* d_row_frames, int batch, int64_t frames,
                              int num_classes, int space_id, int beam_width, float token_min_logp,
                              float beam_prune_logp, const vasr_lm* lm, int32_t* d_ids, int32_t* d_id_len,
                              float* d_score, void* d_ws, size_t ws_bytes, vasr_stream stream) {
  if (!d_logp || !d_ids || !d_id_len || !d_score || !d_ws || batch <= 0 || frames <= 0)
    return fail(VASR_ERR_INVALID, "bad argument");
  if (num_classes < 2 || num_classes > 128 || beam_width < 1 || beam_width > kBeamMax)
    return fail(VASR_ERR_UNSUPPORTED, "beam search supports 2..128 classes and beam_width 1..%d", kBeamMax);
  if (space_id < -1 || space_id >= num_classes - 1) return fail(VASR_ERR_INVALID, "space_id out of range");
  const size_t need_bytes = vasr_beam_workspace_bytes(batch, frames);
  if (ws_bytes < need_bytes) return fail(VASR_ERR_WORKSPACE, "workspace %zu < %zu bytes", ws_bytes, need_bytes);
  // <= 64 utterances: an utterance on four wavefronts of a compute unit (beam_group.hip: the serving latency, and the shorter
  // stay in the way of an overlapped acoustic pass); beyond that
  // one wavefront per utterance, four utterances per compute unit (beam_wave.hip).  Same bits either way; VASR_BEAM_GROUP
  // (devtools build: 0 | 1 = never, 4 = always the four-wavefront form; anything else aborts) pins the form for A/B runs.
  const int e = launch_beam_search_group(
      d_logp, batch, (int)frames, num_classes, space_id < 0 ? 255 : space_id, beam_width, token_min_logp, beam_prune_logp,
      lm ? &lm->view : nullptr, static_cast<unsigned int*>(d_ws), d_ids, d_id_len, d_score, static_cast<hipStream_t>(stream),
      d_row_frames);
  if (e) return fail(VASR_ERR_HIP, "beam search: %s", hipGetErrorString((hipError_t)e));
  return check_launch("beam_search");
}

int vasr_lm_create(const void* h_vocab, int vcap, const void* h_ngram, int ncap, const void* h_trie, int trie_buckets,
                   int order, int bos_id, int eos_id, int unk_id, float alpha, float beta, float unk_offset, vasr_lm** out) {
  if (!h_vocab || !h_ngram || !out || vcap <= 0 || ncap <= 0) return fail(VASR_ERR_INVALID, "bad argument");
  if ((vcap & (vcap - 1)) || (ncap & (ncap - 1)) || vcap < 16 || ncap < 16)
    return fail(VASR_ERR_INVALID, "table capacities must be powers of two >= 16");
  if (h_trie && ((trie_buckets & (trie_buckets - 1)) || trie_buckets < 16))
    return fail(VASR_ERR_INVALID, "the trie's bucket count must be a power of two >= 16");
  if (order < 1 || order > 5) return fail(VASR_ERR_UNSUPPORTED, "n-gram order %d (supported: 1..5)", order);
  auto* lm = new vasr_lm();
  void *p0 = nullptr, *p1 = nullptr, *p2 = nullptr;
  hipError_t e = hipMalloc(&p0, (size_t)vcap * 16);
  if (e == hipSuccess) e = hipMalloc(&p1, (size_t)ncap * 16);
  if (e == hipSuccess && h_trie) e = hipMalloc(&p2, (size_t)trie_buckets * 16);
  if (e == hipSuccess) e = hipMemcpy(p0, h_vocab, (size_t)vcap * 16, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemcpy(p1, h_ngram, (size_t)ncap * 16, hipMemcpyHostToDevice);
  if (e == hipSuccess && h_trie) e = hipMemcpy(p2, h_trie, (size_t)trie_buckets * 16, hipMemcpyHostToDevice);
  lm->allocs = {p0, p1, p2};
  if (e != hipSuccess) {
    vasr_lm_destroy(lm);
    return fail(VASR_ERR_HIP, "uploading the n-gram tables: %s", hipGetErrorString(e));
  }
  lm->view = BeamLm{p0, vcap, p1, ncap, p2, h_trie ? trie_buckets : 0, order, bos_id, eos_id, unk_id, alpha, beta, unk_offset};
  *out = lm;
  return 0;
}

void vasr_lm_destroy(vasr_lm* lm) {
  if (!lm) return;
  for (void* p : lm->allocs) if (p) (void)hipFree(p);
  delete lm;
}

uint64_t vasr_beam_hash_init(void) { return beam_hash_init(); }
uint64_t vasr_beam_hash_step(uint64_t h, uint64_t v) { return beam_hash_step(h, v); }

int vasr_profile_begin(vasr_handle* h) {
  if (!h) return fail(VASR_ERR_INVALID, "null handle");
  for (auto& r : h->prof) { h->ev_pool.push_back(r.a); h->ev_pool.push_back(r.b); }
  h->prof.clear();
  h->profiling = true;
  return 0;
}

int vasr_profile_end(vasr_handle* h, double ms[5], int64_t launches[5], double flops[5], double bytes[5]) {
  if (!h || !ms || !launches) return fail(VASR_ERR_INVALID, "bad argument");
  h->profiling = false;
  for (int i = 0; i < 5; ++i) { ms[i] = 0.0; launches[i] = 0; if (flops) flops[i] = 0.0; if (bytes) bytes[i] = 0.0; }
  for (auto& r : h->prof) {
    HIP_TRY(hipEventSynchronize(r.b));
    float t = 0.f;
    HIP_TRY(hipEventElapsedTime(&t, r.a, r.b));
    ms[r.cls] += t;
    launches[r.cls] += 1;
    if (flops) flops[r.cls] += r.flops;
    if (bytes) bytes[r.cls] += r.bytes;
    h->ev_pool.push_back(r.a);
    h->ev_pool.push_back(r.b);
  }
  h->prof.clear();
  return 0;
}

#ifdef VASR_DEVTOOLS
static __global__ void dev_noop_kernel() {}

int vasr_profile_bracket_overhead(vasr_stream stream, int n, double* out_us) {
  if (n < 1 || n > 4096 || !out_us) return fail(VASR_ERR_INVALID, "bad argument");
  hipStream_t st = static_cast<hipStream_t>(stream);
  std::vector<hipEvent_t> ev(2 * (size_t)n);
  for (auto& e : ev) HIP_TRY(hipEventCreate(&e));
  for (int i = 0; i < n; ++i) {   // same shape as ProfScope: record, launch, record -- back to back on one stream
    HIP_TRY(hipEventRecord(ev[2 * i], st));
    hipLaunchKernelGGL(dev_noop_kernel, dim3(1), dim3(64), 0, st);
    HIP_TRY(hipEventRecord(ev[2 * i + 1], st));
  }
  HIP_TRY(hipEventSynchronize(ev.back()));
  std::vector<float> t(n);
  for (int i = 0; i < n; ++i) HIP_TRY(hipEventElapsedTime(&t[i], ev[2 * i], ev[2 * i + 1]));
  for (auto& e : ev) (void)hipEventDestroy(e);
  std::sort(t.begin(), t.end());
  *out_us = 1e3 * (double)t[n / 2];
  return 0;
}

#endif  // VASR_DEVTOOLS

int vasr_algorithmic_work(const vasr_handle* h, int batch, int64_t samples, double out[5]) {
  if (!h || !h->finalized || !out) return fail(VASR_ERR_INVALID, "bad argument");
  for (int i = 0; i < 5; ++i) out[i] = 0.0;
  int64_t T = vasr_mel_frames(h, samples);
  if (h->has_frontend) {
    const double per_frame = 2.5 * 512 * 9 + 2.0 * 257 + 2.0 * 64 * 23;
    out[4] = per_frame * (double)T * batch;
  }
  int64_t t = T;
  for (const Block& B : h->blocks) {
    const int64_t t_in = t;
    for (const SubBlock& S : B.subs) {
      if (S.separable) {
        const int64_t to = conv_out_frames(t, S.dw);
        out[1] += 2.0 * S.dw.kernel * S.dw.cin * (double)to * batch;
        out[2] += 4.0 * ((double)S.dw.cin * t + (double)S.dw.cin * to) * batch + 4.0 * S.dw.cin * S.dw.kernel;
        t = to;
      }
      out[0] += 2.0 * S.pw.cin * S.pw.cout * (double)t * batch;
    }
    if (B.has_res) out[0] += 2.0 * B.res.cin * B.res.cout * (double)t_in * batch;
  }
  if (h->has_decoder) out[3] = 2.0 * h->dec_feat_in * h->num_classes * (double)t * batch;
  return 0;
}

#ifdef VASR_DEVTOOLS   // ---- include/vasr_devtools.h: isolated layers and weight packers, libvasr_hip_dev.so only ----
int vasr_bench_depthwise(const float* d_x, const float* d_w, const int32_t* d_lens, int batch, int channels,
                         int64_t frames, int kernel, float* d_y, vasr_stream stream) {
  if (!d_x || !d_w || !d_lens || !d_y) return fail(VASR_ERR_INVALID, "bad argument");
  const int64_t ld = pad_frames(frames);
  (void)launch_depthwise(d_x, ld, (int)frames, d_w, d_lens, d_lens, batch, channels, kernel, 1, 1, kernel / 2, d_y, ld,
                         static_cast<hipStream_t>(stream));
  return check_launch("bench_depthwise");
}

int vasr_depthwise_mfma_table_size(int kernel, int dilation) { return depthwise_mfma_table_size(kernel, dilation); }

int vasr_pack_depthwise_taps(const float* h_w, int channels, int kernel, int dilation, uint32_t* h_table, float* h_inv) {
  const int tsz = depthwise_mfma_table_size(kernel, dilation);
  if (!h_w || !h_table || !h_inv || channels <= 0 || !tsz) return fail(VASR_ERR_INVALID, "bad argument / shape not covered");
  for (int c = 0; c < channels; ++c)
    h_inv[c] = pack_depthwise_taps_f16x2(h_w + (size_t)c * kernel, kernel, dilation, tsz, h_table + (size_t)c * tsz);
  return 0;
}

int vasr_bench_depthwise_mfma(const float* d_x, const uint32_t* d_taps, const float* d_tap_inv, const int32_t* d_lens,
                              int batch, int channels, int64_t frames, int kernel, int dilation, float* d_y,
                              uint32_t* d_amax, int amax_stride, vasr_stream stream) {
  if (!d_x || !d_taps || !d_tap_inv || !d_lens || !d_y || !d_amax) return fail(VASR_ERR_INVALID, "bad argument");
  const int64_t ld = pad_frames(frames);
  if (amax_stride < 256 || amax_stride < depthwise_amax_slots(channels, ld)) return fail(VASR_ERR_INVALID, "maxima table too small");
  hipStream_t st = static_cast<hipStream_t>(stream);
  AmaxTab ax{d_amax, amax_stride, 0}, ay{d_amax + (size_t)batch * amax_stride, amax_stride, 0};
  launch_amax(d_x, ld, channels, (int)frames, d_lens, batch, &ax, st);
  const int e = launch_depthwise_mfma(d_x, ld, d_taps, d_tap_inv, d_lens, d_lens, ax, batch, channels, kernel, dilation,
                                      d_y, ld, &ay, st);
  if (e > 0) return fail(VASR_ERR_HIP, "depthwise (MFMA): %s", hipGetErrorString((hipError_t)e));
  if (e < 0) return fail(VASR_ERR_UNSUPPORTED, "no Toeplitz instantiation for kernel %d dilation %d", kernel, dilation);
  if (ay.n < amax_stride)
    HIP_TRY(hipMemset2DAsync(d_amax + (size_t)batch * amax_stride + ay.n, (size_t)amax_stride * 4, 0,
                             (size_t)(amax_stride - ay.n) * 4, batch, st));
  if (ax.n < amax_stride)
    HIP_TRY(hipMemset2DAsync(d_amax + ax.n, (size_t)amax_stride * 4, 0, (size_t)(amax_stride - ax.n) * 4, batch, st));
  return check_launch("bench_depthwise_mfma");
}

int vasr_pack_pointwise(const float* h_w, int cout, int cin, int m_pad, float* h_out) {
  if (!h_w || !h_out || cout <= 0 || cin % 8 || m_pad % 32 || m_pad < cout) return fail(VASR_ERR_INVALID, "bad argument");
  pack_pointwise_weights(h_w, cout, cin, m_pad, h_out);
  return 0;
}

int vasr_bench_pointwise(const float* d_x, const float* d_wt, const float* d_scale, const float* d_shift, int batch,
                         int cin, int cout, int64_t frames, float* d_y, vasr_stream stream) {
  // in_channels % 64 like vasr_load_weight: the K % 32 tile (128 x 256) assumes a 256-frame pitch pad_frames() no longer gives
  if (!d_x || !d_wt || !d_scale || !d_shift || !d_y || cout % 128 || cin % 64)
    return fail(VASR_ERR_INVALID, "bad argument");
  const int64_t ld = pad_frames(frames);
  PwArgs a{};
  a.wt = d_wt; a.x = d_x; a.lens = nullptr; a.scale = d_scale; a.shift = d_shift; a.res = nullptr; a.y = d_y;
  a.M = cout; a.K = cin; a.batch = batch; a.ldx = ld; a.ldy = ld; a.ldr = 0; a.frames = (int)frames;
  a.store_cols = (int)ld; a.m_store = cout; a.relu = 1;
  launch_pointwise(a, static_cast<hipStream_t>(stream));
  return check_launch("bench_pointwise");
}

int vasr_bench_mfma_sustained(int gemm_mode, int workgroups, int steps, float* d_sink, double* flops, vasr_stream stream) {
  if (workgroups < 1 || steps < 1 || !d_sink || !flops) return fail(VASR_ERR_INVALID, "bad argument");
  if (gemm_mode != 1 && gemm_mode != 3) return fail(VASR_ERR_INVALID, "gemm mode %d has no 16-bit MFMA stream (1 = bf16x3, 3 = f16x2)", gemm_mode);
  *flops = launch_mfma_sustained(gemm_mode, workgroups, steps, d_sink, static_cast<hipStream_t>(stream));
  return check_launch("mfma_sustained");
}

int vasr_pack_pointwise_bf16x3(const float* h_w, int cout, int cin, int m_pad, uint16_t* h_out) {
  if (!h_w || !h_out || cout <= 0 || cin % 16 || m_pad % 32 || m_pad < cout) return fail(VASR_ERR_INVALID, "bad argument");
  pack_pointwise_weights_bf16x3(h_w, cout, cin, m_pad, h_out);
  return 0;
}

int vasr_pack_pointwise_f16x2(const float* h_w, int cout, int cin, int m_pad, uint16_t* h_out, float* inv_scale) {
  if (!h_w || !h_out || !inv_scale || cout <= 0 || cin % 16 || m_pad % 32 || m_pad < cout)
    return fail(VASR_ERR_INVALID, "bad argument");
  *inv_scale = pack_pointwise_weights_f16x2(h_w, cout, cin, m_pad, h_out);
  return 0;
}

int vasr_bench_pointwise_f16x2(const float* d_x, const uint16_t* d_w16, float w_inv_scale, const float* d_scale,
                               const float* d_shift, int batch, int cin, int cout, int64_t frames, float* d_y,
                               uint32_t* d_amax, int amax_stride, vasr_stream stream) {
  if (!d_x || !d_w16 || !d_scale || !d_shift || !d_y || !d_amax || !pointwise_split_supported(cout, cin, 0))
    return fail(VASR_ERR_INVALID, "bad argument");
  const int64_t ld = pad_frames(frames);
  if (amax_stride < 256 || amax_stride < pointwise_amax_slots(cout, ld)) return fail(VASR_ERR_INVALID, "maxima table too small");
  hipStream_t st = static_cast<hipStream_t>(stream);
  AmaxTab ax{d_amax, amax_stride, 0};
  launch_amax(d_x, ld, cin, (int)frames, nullptr, batch, &ax, st);
  PwArgs a{};
  a.wt = reinterpret_cast<const float*>(d_w16); a.x = d_x; a.scale = d_scale; a.shift = d_shift; a.y = d_y;
  a.M = cout; a.K = cin; a.batch = batch; a.ldx = ld; a.ldy = ld; a.frames = (int)frames;
  a.store_cols = (int)ld; a.m_store = cout; a.relu = 1;
  a.amax_x = ax; a.w_inv_scale = w_inv_scale;
  a.amax_y = AmaxTab{d_amax + (size_t)batch * amax_stride, amax_stride, 0};   // second table: maxima of y
  int n_y = 0;
  const int e = launch_pointwise_split(a, 2, st, &n_y);
  if (e) return fail(VASR_ERR_HIP, "pointwise GEMM: %s", hipGetErrorString((hipError_t)e));
  // slots past the ones the launch used read as zero for the caller
  if (n_y < amax_stride)
    HIP_TRY(hipMemset2DAsync(d_amax + (size_t)batch * amax_stride + n_y, (size_t)amax_stride * 4, 0,
                             (size_t)(amax_stride - n_y) * 4, batch, st));
  if (ax.n < amax_stride)
    HIP_TRY(hipMemset2DAsync(d_amax + ax.n, (size_t)amax_stride * 4, 0, (size_t)(amax_stride - ax.n) * 4, batch, st));
  return check_launch("bench_pointwise_f16x2");
}

int vasr_bench_pointwise_bf16x3(const float* d_x, const uint16_t* d_w3, const float* d_scale, const float* d_shift,
                                int batch, int cin, int cout, int64_t frames, float* d_y, vasr_stream stream) {
  if (!d_x || !d_w3 || !d_scale || !d_shift || !d_y || !pointwise_split_supported(cout, cin, 0))
    return fail(VASR_ERR_INVALID, "bad argument");
  const int64_t ld = pad_frames(frames);
  PwArgs a{};
  a.wt = reinterpret_cast<const float*>(d_w3); a.x = d_x; a.scale = d_scale; a.shift = d_shift; a.y = d_y;
  a.M = cout; a.K = cin; a.batch = batch; a.ldx = ld; a.ldy = ld; a.frames = (int)frames;
  a.store_cols = (int)ld; a.m_store = cout; a.relu = 1;
  const int e = launch_pointwise_split(a, 0, static_cast<hipStream_t>(stream));
  if (e) return fail(VASR_ERR_HIP, "pointwise GEMM: %s", hipGetErrorString((hipError_t)e));
  return check_launch("bench_pointwise_bf16x3");
}

#endif  // VASR_DEVTOOLS

}  // extern "C"
