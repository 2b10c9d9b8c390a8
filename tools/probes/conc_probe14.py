"""conc_probe13 WITHOUT torch in the process: pure ctypes on the SYSTEM HIP runtime (/opt/rocm), the pre-fix library variant, the synthetic
attacker, inputs from tools/probes/stft_mfma_repro_dump.py (run beforehand in another process).  Old-ABI FrontendDesc (commit 25e1455)."""
import ctypes as C, os, sys, threading, time
import numpy as np
assert "torch" not in sys.modules
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); d = sys.argv[1]
hip = C.CDLL("/opt/rocm/lib/libamdhip64.so", mode=C.RTLD_GLOBAL)
hip.hipMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]; hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
hip.hipStreamCreateWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_uint]; hip.hipStreamSynchronize.argtypes = [C.c_void_p]
class FrontendDesc(C.Structure):
    _fields_ = [("sample_rate", C.c_int32), ("n_fft", C.c_int32), ("win_length", C.c_int32), ("hop_length", C.c_int32), ("n_mels", C.c_int32),
                ("preemph", C.c_float), ("log_guard", C.c_float), ("normalize", C.c_int32), ("h_window", C.POINTER(C.c_float)), ("h_filterbank", C.POINTER(C.c_float))]
class ModelDesc(C.Structure):
    _fields_ = [("frontend", C.POINTER(FrontendDesc)), ("feat_in", C.c_int32), ("n_blocks", C.c_int32), ("blocks", C.c_void_p), ("dec_feat_in", C.c_int32), ("num_classes", C.c_int32)]
L = C.CDLL(os.path.join(R, "viet-asr_amd", "lib", "var_old_frontend.so"))
L.vasr_create.argtypes = [C.POINTER(ModelDesc), C.POINTER(C.c_void_p)]; L.vasr_finalize.argtypes = [C.c_void_p]
L.vasr_melspec_f32.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]
win = np.fromfile(os.path.join(d, "win.bin"), dtype=np.float32); fb = np.fromfile(os.path.join(d, "fb.bin"), dtype=np.float32); sig = np.fromfile(os.path.join(d, "wav.bin"), dtype=np.float32)
fe = FrontendDesc(16000, 512, 320, 160, 64, 0.97, 2.0 ** -24, 0, win.ctypes.data_as(C.POINTER(C.c_float)), fb.ctypes.data_as(C.POINTER(C.c_float)))
md = ModelDesc(C.pointer(fe), 64, 0, None, 0, 0)
h = C.c_void_p(); assert L.vasr_create(C.byref(md), C.byref(h)) == 0; assert L.vasr_finalize(h) == 0
def dmalloc(n):
    p = C.c_void_p(); assert hip.hipMalloc(C.byref(p), n) == 0; return p
def stream():
    s = C.c_void_p(); assert hip.hipStreamCreateWithFlags(C.byref(s), 1) == 0; return s
B, Ls = 64, 160000; T = 1 + Ls // 160
lens = np.full(B, Ls, dtype=np.int64)
d_wav, d_len, d_mel, d_seq = dmalloc(sig.nbytes), dmalloc(8 * B), dmalloc(4 * B * 64 * T), dmalloc(8 * B)
hip.hipMemcpy(d_wav, sig.ctypes.data_as(C.c_void_p), sig.nbytes, 1); hip.hipMemcpy(d_len, lens.ctypes.data_as(C.c_void_p), 8 * B, 1)
sa, sb = stream(), stream()
P = C.CDLL(os.path.join(R, "viet-asr_amd", "lib", "probe_mfma_attacker.so"))
P.mfma_attacker_launch.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
d_sink = dmalloc(1 << 20)
def victim():
    assert L.vasr_melspec_f32(h, d_wav, d_len, B, Ls, d_mel, d_seq, sa) == 0; hip.hipStreamSynchronize(sa)
    out = np.empty((B, 64, T), dtype=np.float32); hip.hipMemcpy(out.ctypes.data_as(C.c_void_p), d_mel, out.nbytes, 2); return out
want = victim()
for attack in (False, True):
    stop = [False]; bad = 0; n = 0
    def b():
        while attack and not stop[0]:
            P.mfma_attacker_launch(2048, 24576, 600, 0, d_sink, None, sb); hip.hipStreamSynchronize(sb)
    tb = threading.Thread(target=b); tb.start(); t0 = time.time()
    while time.time() - t0 < 6:
        out = victim(); n += 1; bad += int(not np.array_equal(out.view(np.uint32), want.view(np.uint32)))
    stop[0] = True; tb.join()
    print(f"pure ctypes, NO torch in the process, system runtime | {'synthetic MFMA attacker' if attack else 'idle device':24s}: calls {n} wrong {bad}", flush=True)
