"""Does torch matter?  The pre-fix front end (library variant) called through the C ABI on RAW HIP streams and hipMalloc'ed buffers (ctypes on
libamdhip64: no torch tensor, no torch stream touches the victim or the attacker), the synthetic MFMA attacker on a second raw stream, two
host threads.  torch is imported (the package needs it to load) but idle."""
import ctypes as C, os, sys, threading, time
import numpy as np
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, R)
import viet_asr_amd
from viet_asr_amd import configs, synth, _lib
from viet_asr_amd.frontend_tables import frontend_description
hip = C.CDLL("libamdhip64.so")
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]; hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
hip.hipStreamCreateWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_uint]; hip.hipStreamSynchronize.argtypes = [C.c_void_p]
def dmalloc(n):
    p = C.c_void_p(); assert hip.hipMalloc(C.byref(p), n) == 0; return p
def stream():
    s = C.c_void_p(); assert hip.hipStreamCreateWithFlags(C.byref(s), 1) == 0; return s     # hipStreamNonBlocking
cfg = configs.builtin("quartznet12x1_vi")
h = _lib.Handle(frontend=frontend_description(dict(cfg["AudioToMelSpectrogramPreprocessor"], normalize=None))); h.finalize()
L = _lib.lib()
B, Ls = 64, 160000
sig, lens = synth.audio_batch(B, Ls, 3, ragged=False)
T = 1 + Ls // 160
d_wav, d_len, d_mel, d_seq = dmalloc(sig.nbytes), dmalloc(8 * B), dmalloc(4 * B * 64 * T), dmalloc(8 * B)
hip.hipMemcpy(d_wav, sig.ctypes.data_as(C.c_void_p), sig.nbytes, 1); hip.hipMemcpy(d_len, lens.ctypes.data_as(C.c_void_p), 8 * B, 1)
sa, sb = stream(), stream()
P = C.CDLL(os.path.join(R, "viet-asr_amd", "lib", "probe_mfma_attacker.so"))
P.mfma_attacker_launch.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
d_sink = dmalloc(1 << 20)
def victim():
    _lib.check(L.vasr_melspec_f32(h.h, d_wav, d_len, B, Ls, d_mel, d_seq, sa)); hip.hipStreamSynchronize(sa)
    out = np.empty((B, 64, T), dtype=np.float32); hip.hipMemcpy(out.ctypes.data_as(C.c_void_p), d_mel, out.nbytes, 2); return out
want = victim()
for attack in (False, True):
    stop = [False]; bad = 0; n = 0
    def b():
        while attack and not stop[0]:
            P.mfma_attacker_launch(2048, 24576, 600, 0, d_sink, None, sb); hip.hipStreamSynchronize(sb)
    tb = threading.Thread(target=b); tb.start()
    t0 = time.time()
    while time.time() - t0 < 6:
        out = victim(); n += 1; bad += int(not np.array_equal(out.view(np.uint32), want.view(np.uint32)))
    stop[0] = True; tb.join()
    print(f"[{os.path.basename(_lib.LIB_PATH)}] raw HIP streams + hipMalloc, no torch object involved | {'synthetic MFMA attacker' if attack else 'idle device':24s}: calls {n} wrong {bad}", flush=True)
