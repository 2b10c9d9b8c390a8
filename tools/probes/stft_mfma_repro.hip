// Stand-alone reproducer of profiles/r06_concurrency.txt: this library's STFT / log-mel kernel (csrc/frontend.hip, linked in as it is) next
// to a synthetic 16-bit MFMA kernel on a second stream.  No torch, no libvasr, no Python: two kernels, two streams, one process.
//   bash tools/probes/stft_mfma_repro.sh        builds it TWICE -- frontend.hip with the SLP vectoriser (packed-FP32 instructions, how the
//                                               kernel was built until round 6) and with -fno-slp-vectorize (how it ships now) -- and runs both
// Expected on MI355X: with packed-FP32 instructions a large share of the launches next to the attacker return wrong values, none on the
// idle device; without them none in either case.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <atomic>
#include <chrono>
#include <cstdlib>
#include <thread>
#include <vector>
#include "vasr_internal.h"
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f32x16 = __attribute__((ext_vector_type(16))) float;
#ifndef MODE_NORMALIZE
#define MODE_NORMALIZE 0
#endif
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 2; } } while (0)

__global__ __launch_bounds__(128, 2) void attacker(float* sink, int iters) {
  extern __shared__ unsigned char smem[];
  f16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(0.01f * (threadIdx.x + i)); b[i] = (_Float16)(0.02f * (i + 1)); }
  f32x16 acc0 = {}, acc1 = {};
  for (int it = 0; it < iters; ++it) {
    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, acc1, 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
  if (s == 12345.678f) sink[0] = s + smem[threadIdx.x];
}

template <class T> static T* up(const std::vector<T>& v) { T* d = nullptr; hipMalloc(&d, v.size() * sizeof(T)); hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice); return d; }

int main(int argc, char** argv) {
  const int pace_us = argc > 1 ? atoi(argv[1]) : 0;
  const int B = 64, L = 160000, hop = 160, T = 1 + L / hop;
  // front-end tables as vasr_api.cpp build_frontend makes them: hann(320) centred in 512, the two twiddle tables, 64 triangular filters
  std::vector<float> win(512, 0.f), tw256(512), tw512(2 * 258, 0.f), mw(64 * vasr::kMelTaps, 0.f);
  std::vector<int32_t> lo(64);
  for (int i = 0; i < 320; ++i) win[96 + i] = (float)(0.5 - 0.5 * std::cos(2.0 * M_PI * i / 319.0));
  for (int m = 0; m < 256; ++m) { tw256[2 * m] = (float)std::cos(-2.0 * M_PI * m / 256.0); tw256[2 * m + 1] = (float)std::sin(-2.0 * M_PI * m / 256.0); }
  for (int k = 0; k <= 256; ++k) { tw512[2 * k] = (float)std::cos(-2.0 * M_PI * k / 512.0); tw512[2 * k + 1] = (float)std::sin(-2.0 * M_PI * k / 512.0); }
  for (int f = 0; f < 64; ++f) { lo[f] = 3 * f; for (int i = 0; i < 8 + f / 4; ++i) mw[f * vasr::kMelTaps + i] = 0.02f * (1.f - std::fabs((i - (4 + f / 8)) / (5.f + f / 8))); }
  std::vector<float> wav((size_t)B * L);
  unsigned s = 12345u;
  for (auto& v : wav) { s = s * 1664525u + 1013904223u; v = ((int)(s >> 9) - (1 << 22)) * (0.1f / (1 << 22)); }
  // optional: the library's REAL inputs, dumped by tools/probes/stft_mfma_repro_dump.py -- <dir>/wav.bin [64][160000] f32, win.bin [320] f32,
  // fb.bin [64][257] f32 (tables then built as vasr_api.cpp build_frontend builds them)
  if (argc > 3) {
    const std::string d = argv[3];
    auto rd = [&](const char* name, float* dst, size_t n) { FILE* f = fopen((d + "/" + name).c_str(), "rb"); if (!f || fread(dst, 4, n, f) != n) { printf("cannot read %s\n", name); exit(3); } fclose(f); };
    std::vector<float> w320(320), fb(64 * 257);
    rd("wav.bin", wav.data(), wav.size()); rd("win.bin", w320.data(), 320); rd("fb.bin", fb.data(), fb.size());
    std::fill(win.begin(), win.end(), 0.f); std::fill(mw.begin(), mw.end(), 0.f);
    for (int i = 0; i < 320; ++i) win[96 + i] = w320[i];
    for (int f = 0; f < 64; ++f) {
      int first = -1, last = -1;
      for (int k = 0; k < 257; ++k) if (fb[f * 257 + k] != 0.f) { if (first < 0) first = k; last = k; }
      lo[f] = first < 0 ? 0 : first;
      for (int k = first; first >= 0 && k <= last; ++k) mw[f * vasr::kMelTaps + (k - first)] = fb[f * 257 + k];
    }
    printf("(real inputs from %s)\n", d.c_str());
  }
  vasr::FrontendTables tb{up(win), up(tw256), up(tw512), up(mw), up(lo), 0};
  float* d_wav = up(wav);
  float *d_mel, *d_sink;
  CK(hipMalloc(&d_mel, (size_t)B * 64 * T * 4)); CK(hipMalloc(&d_sink, 4096));
  // (HIP deals streams to a few hardware queues in turn: several are created and two far apart are used, so that victim and attacker do
  //  not end up serialised behind each other in one queue)
  hipStream_t pool[8];
  for (auto& q : pool) CK(hipStreamCreateWithFlags(&q, hipStreamNonBlocking));
  hipStream_t sa = pool[1], sb = pool[2 + (argc > 2 ? atoi(argv[2]) : 0)];
  std::vector<float> want((size_t)B * 64 * T), got(want.size());
  // the three launches of vasr_melspec_f32 (normalisation off: the third one only masks the frames past each row's length)
  std::vector<int64_t> lens(B, L);
  int64_t* d_len = up(lens);
  int64_t* d_seq = nullptr;
  CK(hipMalloc(&d_seq, B * 8));
  auto melspec = [&]() {
    vasr::launch_seq_len(d_len, B, hop, d_seq, sa);
    vasr::launch_stft_logmel(tb, d_wav, false, B, L, nullptr, hop, 0.97f, 5.9604645e-8f, d_mel, T, T, sa);
    vasr::launch_normalize(d_mel, T, d_seq, B, 64, T, MODE_NORMALIZE, sa);
  };
  melspec();
  CK(hipStreamSynchronize(sa));
  CK(hipMemcpy(want.data(), d_mel, want.size() * 4, hipMemcpyDeviceToHost));
  for (int phase = 0; phase < 2; ++phase) {
    int bad_launches = 0; long bad_values = 0; const int launches = 400;
    std::atomic<bool> stop{false};
    // the attacker runs from a host thread of its own, launch + synchronise in a loop, as a second client of the GPU would
    std::thread other([&] {
      while (phase && !stop.load()) {
        hipLaunchKernelGGL(attacker, dim3(2048), dim3(128), 24 * 1024, sb, d_sink, 600);
        (void)hipStreamSynchronize(sb);
        if (pace_us) std::this_thread::sleep_for(std::chrono::microseconds(pace_us));   // a Python client's pace between launches
      }
    });
    double dev_us = 0;
    for (int l = 0; l < launches; ++l) {
      const auto t0 = std::chrono::steady_clock::now();
      melspec();
      CK(hipStreamSynchronize(sa));
      dev_us += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
      CK(hipMemcpy(got.data(), d_mel, got.size() * 4, hipMemcpyDeviceToHost));
      long n = 0;
      for (size_t i = 0; i < got.size(); ++i) n += memcmp(&got[i], &want[i], 4) != 0;
      bad_launches += n != 0; bad_values += n;
    }
    stop.store(true);
    other.join();
    CK(hipStreamSynchronize(sb));
    printf("seq_len + stft_logmel + mask kernels, 64 x 10 s, %-32s: %d launches, %d with a wrong value (%ld of %zu values per launch on average), %.0f us per victim call\n",
           phase ? "next to the f16 MFMA kernel" : "idle device", launches, bad_launches, bad_launches ? bad_values / bad_launches : 0, got.size(), dev_us / launches);
  }
  return 0;
}
