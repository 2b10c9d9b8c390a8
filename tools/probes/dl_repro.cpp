// conc_probe14.py in C++: dlopen()s a build of the library and the attacker .so, reads the dumped inputs, runs victim + attacker from two
// host threads.   g++ -O2 -std=c++17 -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ dl_repro.cpp -L/opt/rocm/lib -lamdhip64 -ldl -lpthread
//   ./dl_repro <libvasr .so> <attacker .so> <input dir> [normalize 0|1] [attacker LDS bytes] [old_abi]
#include <hip/hip_runtime_api.h>
#include <dlfcn.h>
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
struct FrontendDescOld { int32_t sample_rate, n_fft, win_length, hop_length, n_mels; float preemph, log_guard; int32_t normalize; const float* h_window; const float* h_filterbank; };
struct FrontendDescNew { int32_t sample_rate, n_fft, win_length, hop_length, n_mels; float preemph, log_guard; int32_t normalize; const float* h_window; const float* h_filterbank; int32_t log_guard_clamp; };
struct ModelDesc { const void* frontend; int32_t feat_in, n_blocks; const void* blocks; int32_t dec_feat_in, num_classes; };
static std::vector<float> rd(const std::string& p, size_t n) { std::vector<float> v(n); FILE* f = fopen(p.c_str(), "rb"); if (!f || fread(v.data(), 4, n, f) != n) { printf("cannot read %s\n", p.c_str()); exit(3); } fclose(f); return v; }
int main(int argc, char** argv) {
  void* L = dlopen(argv[1], RTLD_NOW | RTLD_GLOBAL); void* A = dlopen(argv[2], RTLD_NOW | RTLD_GLOBAL);
  if (!L || !A) { printf("dlopen: %s\n", dlerror()); return 2; }
  const std::string d = argv[3]; const int normalize = argc > 4 ? atoi(argv[4]) : 0, att_lds = argc > 5 ? atoi(argv[5]) : 24576; const bool old_abi = argc > 6;
  auto create = (int (*)(const ModelDesc*, void**))dlsym(L, "vasr_create"); auto fin = (int (*)(void*))dlsym(L, "vasr_finalize");
  auto mel = (int (*)(void*, const float*, const int64_t*, int, int64_t, float*, int64_t*, void*))dlsym(L, "vasr_melspec_f32");
  auto att = (int (*)(int, int, int, int, float*, const float*, void*))dlsym(A, "mfma_attacker_launch");
  const int B = 64, Ls = 160000, T = 1 + Ls / 160;
  auto win = rd(d + "/win.bin", 320), fb = rd(d + "/fb.bin", 64 * 257), wav = rd(d + "/wav.bin", (size_t)B * Ls);
  FrontendDescOld fo{16000, 512, 320, 160, 64, 0.97f, 5.9604645e-8f, normalize, win.data(), fb.data()};
  FrontendDescNew fn{16000, 512, 320, 160, 64, 0.97f, 5.9604645e-8f, normalize, win.data(), fb.data(), 0};
  ModelDesc md{old_abi ? (const void*)&fo : (const void*)&fn, 64, 0, nullptr, 0, 0};
  void* h = nullptr; if (create(&md, &h) || fin(h)) { printf("create failed\n"); return 2; }
  float *d_wav, *d_mel, *d_sink; int64_t *d_len, *d_seq; std::vector<int64_t> lens(B, Ls);
  hipMalloc((void**)&d_wav, wav.size() * 4); hipMalloc((void**)&d_mel, (size_t)B * 64 * T * 4); hipMalloc((void**)&d_sink, 1 << 20); hipMalloc((void**)&d_len, B * 8); hipMalloc((void**)&d_seq, B * 8);
  hipMemcpy(d_wav, wav.data(), wav.size() * 4, hipMemcpyHostToDevice); hipMemcpy(d_len, lens.data(), B * 8, hipMemcpyHostToDevice);
  hipStream_t sa, sb; hipStreamCreateWithFlags(&sa, hipStreamNonBlocking); hipStreamCreateWithFlags(&sb, hipStreamNonBlocking);
  std::vector<float> want((size_t)B * 64 * T), got(want.size());
  mel(h, d_wav, d_len, B, Ls, d_mel, d_seq, sa); hipStreamSynchronize(sa); hipMemcpy(want.data(), d_mel, want.size() * 4, hipMemcpyDeviceToHost);
  for (int phase = 0; phase < 2; ++phase) {
    std::atomic<bool> stop{false}; int bad = 0; const int n = 400;
    std::thread other([&] { while (phase && !stop.load()) { att(2048, att_lds, 600, 0, d_sink, nullptr, sb); hipStreamSynchronize(sb); } });
    for (int l = 0; l < n; ++l) {
      mel(h, d_wav, d_len, B, Ls, d_mel, d_seq, sa); hipStreamSynchronize(sa); hipMemcpy(got.data(), d_mel, got.size() * 4, hipMemcpyDeviceToHost);
      bad += memcmp(got.data(), want.data(), got.size() * 4) != 0;
    }
    stop.store(true); other.join();
    printf("dlopen(%s), normalize %d, attacker LDS %5d | %-24s: calls %d wrong %d\n", strrchr(argv[1], '/') + 1, normalize, att_lds, phase ? "synthetic MFMA attacker" : "idle device", n, bad);
  }
  return 0;
}
