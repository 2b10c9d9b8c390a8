"""ctypes binding of libvasr_hip.so (include/vasr.h) and, for tests / tools, of libvasr_hip_dev.so (vasr_devtools.h).

The library is the product: there is NO CPU fallback.  If the shared object is missing, or a
compute entry point is called without a HIP device, this module raises -- it never routes
through torch eager ops or the test oracle.
"""
import ctypes as C
import os

import numpy as np
import torch  # noqa: F401  -- must be imported BEFORE the .so: torch ships its own HIP runtime, and the
#                library has to bind to that already-loaded copy (loading /opt/rocm's first leaves two
#                runtimes in the process and hipMalloc then reports "no ROCm-capable device")

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VASR_LIB_PATH") or os.path.join(_HERE, "lib", "libvasr_hip.so")   # override: dev builds
# the -DVASR_DEVTOOLS build: + include/vasr_devtools.h, and the only build that reads the VASR_* kernel-selection switches
DEV_LIB_PATH = os.environ.get("VASR_LIB_PATH") or os.path.join(_HERE, "lib", "libvasr_hip_dev.so")


class VasrError(RuntimeError):
    pass


class BlockDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in
                ("filters", "repeat", "kernel", "stride", "dilation", "residual", "separable")]


class FrontendDesc(C.Structure):
    _fields_ = [("sample_rate", C.c_int32), ("n_fft", C.c_int32), ("win_length", C.c_int32),
                ("hop_length", C.c_int32), ("n_mels", C.c_int32), ("preemph", C.c_float),
                ("log_guard", C.c_float), ("normalize", C.c_int32),
                ("h_window", C.POINTER(C.c_float)), ("h_filterbank", C.POINTER(C.c_float)),
                ("log_guard_clamp", C.c_int32)]


class ModelDesc(C.Structure):
    _fields_ = [("frontend", C.POINTER(FrontendDesc)), ("feat_in", C.c_int32), ("n_blocks", C.c_int32),
                ("blocks", C.POINTER(BlockDesc)), ("dec_feat_in", C.c_int32), ("num_classes", C.c_int32)]


# name -> (restype, argtypes); every symbol include/vasr.h declares
_P = C.c_void_p
SIGNATURES = {
    "vasr_create": (C.c_int, [C.POINTER(ModelDesc), C.POINTER(_P)]),
    "vasr_destroy": (None, [_P]),
    "vasr_load_weight": (C.c_int, [_P, C.c_char_p, _P, C.POINTER(C.c_int64), C.c_int]),
    "vasr_finalize": (C.c_int, [_P]),
    "vasr_mel_frames": (C.c_int64, [_P, C.c_int64]),
    "vasr_encoded_frames": (C.c_int64, [_P, C.c_int64]),
    "vasr_workspace_bytes": (C.c_size_t, [_P, C.c_int, C.c_int64, C.c_int64]),
    "vasr_melspec_f32": (C.c_int, [_P, _P, _P, C.c_int, C.c_int64, _P, _P, _P]),
    "vasr_encoder_f32": (C.c_int, [_P, _P, _P, C.c_int, C.c_int64, _P, _P, _P, C.c_size_t, _P]),
    "vasr_decoder_logsoftmax_f32": (C.c_int, [_P, _P, C.c_int, C.c_int64, _P, _P, C.c_size_t, _P]),
    "vasr_greedy_argmax": (C.c_int, [_P, C.c_int, C.c_int64, C.c_int, _P, _P]),
    "vasr_ctc_collapse": (C.c_int, [_P, C.c_int, C.c_int64, C.c_int, _P, _P, _P]),
    "vasr_transcribe_greedy_f32": (C.c_int, [_P, _P, _P, C.c_int, C.c_int64, _P, _P, _P, _P, _P, _P,
                                             C.c_size_t, _P]),
    "vasr_transcribe_greedy_pcm16": (C.c_int, [_P, _P, _P, C.c_int, C.c_int64, _P, _P, _P, _P, _P, _P,
                                               C.c_size_t, _P]),
    "vasr_pcm16_to_f32": (C.c_int, [_P, C.c_int64, _P, _P]),
    "vasr_resample_f32": (C.c_int, [_P, C.c_int64, _P, C.c_int, _P, C.c_int, C.c_int, C.c_double, _P, C.c_int64, _P, _P]),
    "vasr_set_gemm_mode": (C.c_int, [_P, C.c_int]),
    "vasr_get_gemm_mode": (C.c_int, [_P]),
    "vasr_set_slices": (C.c_int, [_P, C.c_int]),
    "vasr_set_row_independent": (C.c_int, [_P, C.c_int]),
    "vasr_set_busy_cus": (C.c_int, [_P, C.c_int]),
    "vasr_beam_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int64]),
    "vasr_beam_search_f32": (C.c_int, [_P, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, _P,
                                       _P, _P, _P, _P, C.c_size_t, _P]),
    "vasr_beam_search_rows_f32": (C.c_int, [_P, _P, C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float,
                                            _P, _P, _P, _P, _P, C.c_size_t, _P]),
    "vasr_lm_create": (C.c_int, [_P, C.c_int, _P, C.c_int, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float,
                                 C.c_float, C.c_float, C.POINTER(_P)]),
    "vasr_lm_destroy": (None, [_P]),
    "vasr_beam_workgroups": (C.c_int, [C.c_int]),
    "vasr_beam_hash_init": (C.c_uint64, []),
    "vasr_beam_hash_step": (C.c_uint64, [C.c_uint64, C.c_uint64]),
    "vasr_last_error": (C.c_char_p, []),
    "vasr_version": (C.c_char_p, []),
    "vasr_abi_version": (C.c_int, []),
    "vasr_algorithmic_work": (C.c_int, [_P, C.c_int, C.c_int64, C.POINTER(C.c_double)]),
    "vasr_profile_begin": (C.c_int, [_P]),
    "vasr_profile_end": (C.c_int, [_P, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "vasr_padded_frames": (C.c_int64, [C.c_int64]),
}

# include/vasr_devtools.h: exported by libvasr_hip_dev.so only (isolated layers, weight packers, bracket overhead)
DEV_SIGNATURES = {
    "vasr_pack_pointwise_f16x2": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, _P, C.POINTER(C.c_float)]),
    "vasr_bench_pointwise_f16x2": (C.c_int, [_P, _P, C.c_float, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int64, _P, _P,
                                             C.c_int, _P]),
    "vasr_depthwise_mfma_table_size": (C.c_int, [C.c_int, C.c_int]),
    "vasr_pack_depthwise_taps": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, _P, _P]),
    "vasr_bench_depthwise_mfma": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, C.c_int64, C.c_int, C.c_int, _P, _P, C.c_int, _P]),
    "vasr_profile_bracket_overhead": (C.c_int, [_P, C.c_int, C.POINTER(C.c_double)]),
    "vasr_fused_tile_choice": (C.c_int, [C.c_int64, C.c_int]),
    "vasr_pack_pointwise": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, _P]),
    "vasr_pack_pointwise_bf16x3": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, _P]),
    "vasr_bench_pointwise_bf16x3": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int64, _P, _P]),
    "vasr_bench_mfma_sustained": (C.c_int, [C.c_int, C.c_int, C.c_int, _P, C.POINTER(C.c_double), _P]),
    "vasr_bench_depthwise": (C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int64, C.c_int, _P, _P]),
    "vasr_bench_pointwise": (C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int64, _P, _P]),
}

ABI_VERSION = 7          # VASR_ABI_VERSION of the include/vasr.h these signatures were written against

_lib = None
_dev = None


def _load(path, tables):
    if not os.path.exists(path):
        raise VasrError(
            f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C viet-asr_amd/csrc`). There is no CPU fallback for this path.")
    l = C.CDLL(path)
    for table, required in tables:
        for name, (res, args) in table.items():
            fn = getattr(l, name, None)
            if fn is None:
                if required:
                    raise VasrError(f"{path} does not export {name}")
                continue
            fn.restype, fn.argtypes = res, args
    have = l.vasr_abi_version() if hasattr(l, "vasr_abi_version") else 0
    if have != ABI_VERSION:
        raise VasrError(f"{path} implements ABI {have}, these bindings expect {ABI_VERSION}: rebuild it (make -C viet-asr_amd/csrc)")
    return l


def lib():
    """The product library, loaded once; raises loudly when it has not been built.  (A VASR_LIB_PATH override may name a
    devtools build: its extra symbols get their signatures too.)"""
    global _lib
    if _lib is None:
        _lib = _load(LIB_PATH, ((SIGNATURES, True), (DEV_SIGNATURES, False)))
    return _lib


def dev_lib():
    """libvasr_hip_dev.so (tests and tools only): everything the product library has + include/vasr_devtools.h."""
    global _dev
    if _dev is None:
        _dev = lib() if DEV_LIB_PATH == LIB_PATH else _load(DEV_LIB_PATH, ((SIGNATURES, True), (DEV_SIGNATURES, True)))
    return _dev


_ERR_TYPES = {-1: ValueError, -5: NotImplementedError}


def check(rc, l=None):
    """l: the library the failing call went to (its error string lives there); default the product library."""
    if rc != 0:
        msg = (l or lib()).vasr_last_error().decode("utf-8", "replace")
        raise _ERR_TYPES.get(rc, VasrError)(f"libvasr_hip: {msg} (status {rc})")


def _fptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


class Handle:
    """Owns one vasr_handle: any subset of {front end, encoder, CTC head}."""

    def __init__(self, frontend=None, feat_in=0, blocks=None, dec_feat_in=0, num_classes=0):
        L = lib()
        self._keep = []
        md = ModelDesc()
        if frontend is not None:
            fe = FrontendDesc()
            win = np.ascontiguousarray(frontend["window"], dtype=np.float32)
            fb = np.ascontiguousarray(frontend["filterbank"], dtype=np.float32)
            self._keep += [win, fb, fe]
            fe.sample_rate, fe.n_fft = int(frontend["sample_rate"]), int(frontend["n_fft"])
            fe.win_length, fe.hop_length = int(frontend["win_length"]), int(frontend["hop_length"])
            fe.n_mels = int(frontend["n_mels"])
            pre = frontend.get("preemph", 0.97)
            fe.preemph = -1.0 if pre is None else float(pre)
            fe.log_guard = float(frontend.get("log_guard", 2 ** -24))
            fe.normalize = {"per_feature": 1, "all_features": 2}.get(frontend.get("normalize", "per_feature"), 0)
            fe.log_guard_clamp = 1 if frontend.get("log_guard_type", "add") == "clamp" else 0
            if win.shape != (fe.win_length,) or fb.shape != (fe.n_mels, fe.n_fft // 2 + 1):
                raise ValueError(f"window {win.shape} / filterbank {fb.shape} do not match the description")
            fe.h_window, fe.h_filterbank = _fptr(win), _fptr(fb)
            md.frontend = C.pointer(fe)
        blocks = blocks or []
        if blocks:
            arr = (BlockDesc * len(blocks))()
            for i, b in enumerate(blocks):
                arr[i] = BlockDesc(*[int(b[k]) for k in
                                     ("filters", "repeat", "kernel", "stride", "dilation", "residual", "separable")])
            self._keep.append(arr)
            md.blocks, md.n_blocks, md.feat_in = arr, len(blocks), int(feat_in)
        md.dec_feat_in, md.num_classes = int(dec_feat_in), int(num_classes)
        h = _P()
        check(L.vasr_create(C.byref(md), C.byref(h)))
        self.h = h
        self.num_classes = int(num_classes)

    def load_state_dict(self, sd):
        """sd: {reference state_dict key: array-like}; integer tensors (num_batches_tracked) are skipped."""
        L = lib()
        for k, v in sd.items():
            a = np.asarray(v.detach().cpu().numpy() if hasattr(v, "detach") else v)
            if a.dtype.kind != "f":
                continue
            a = np.ascontiguousarray(a, dtype=np.float32)
            shape = (C.c_int64 * max(a.ndim, 1))(*a.shape)
            check(L.vasr_load_weight(self.h, k.encode(), a.ctypes.data_as(_P), shape, a.ndim))

    def finalize(self):
        check(lib().vasr_finalize(self.h))

    def mel_frames(self, samples):
        return int(lib().vasr_mel_frames(self.h, int(samples)))

    def encoded_frames(self, mel_frames):
        return int(lib().vasr_encoded_frames(self.h, int(mel_frames)))

    def workspace_bytes(self, batch, samples=0, mel_frames=0):
        return int(lib().vasr_workspace_bytes(self.h, int(batch), int(samples), int(mel_frames)))

    def algorithmic_work(self, batch, samples):
        out = (C.c_double * 5)()
        check(lib().vasr_algorithmic_work(self.h, int(batch), int(samples), out))
        return dict(pointwise_flops=out[0], depthwise_flops=out[1], depthwise_bytes=out[2],
                    decoder_flops=out[3], frontend_flops=out[4])

    GEMM_MODES = {"fp32": 0, "bf16x3": 1, "bf16x2": 2, "f16x2": 3}

    def set_gemm_mode(self, mode):
        """'fp32' (exact fp32 MFMA), 'bf16x3' (3 x bf16 split operands), 'f16x2' (2 x fp16 scaled split operands) --
        all fp32-equivalent accuracy -- or the reduced-precision opt-in 'bf16x2'; see vasr_set_gemm_mode."""
        check(lib().vasr_set_gemm_mode(self.h, self.GEMM_MODES[mode] if isinstance(mode, str) else int(mode)))

    def gemm_mode_name(self):
        m = int(lib().vasr_get_gemm_mode(self.h))
        return {v: k for k, v in self.GEMM_MODES.items()}[m]

    def set_slices(self, n):
        check(lib().vasr_set_slices(self.h, int(n)))

    def set_row_independent(self, on):
        check(lib().vasr_set_row_independent(self.h, int(bool(on))))

    def set_busy_cus(self, cus):
        check(lib().vasr_set_busy_cus(self.h, int(cus)))

    def profile_begin(self):
        check(lib().vasr_profile_begin(self.h))

    def profile_end(self):
        ms, n, fl, by = (C.c_double * 5)(), (C.c_int64 * 5)(), (C.c_double * 5)(), (C.c_double * 5)()
        check(lib().vasr_profile_end(self.h, ms, n, fl, by))
        names = ("frontend", "depthwise", "pointwise", "head", "fused")
        return {k: dict(ms=ms[i], launches=int(n[i]), flops=fl[i], bytes=by[i]) for i, k in enumerate(names)}

    @staticmethod
    def profile_bracket_overhead_us(stream, n=256):
        out = C.c_double()
        check(dev_lib().vasr_profile_bracket_overhead(stream, n, C.byref(out)), dev_lib())
        return out.value

    def close(self):
        if getattr(self, "h", None):
            lib().vasr_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
