"""CPU-side checks: the C-ABI library loads and exports every symbol include/vasr.h declares, argument
validation works without a GPU, and the host-side NeuralModule mirror behaves like the reference's."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT
from viet_asr_amd import _lib, configs, synth
from viet_asr_amd import asr as nemo_asr
from viet_asr_amd.core import (DeviceType, NeuralModuleFactory, NeuralPortNameMismatchError,
                               NeuralPortNmTensorMismatchError, NmTensor)


def _header_functions(name="vasr.h"):
    src = open(os.path.join(ROOT, "include", name)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(vasr_[a-z0-9_]+)\s*\(", src)))


def _dynamic_symbols(path):
    """Defined symbols of the shared object's dynamic table, read from the ELF itself (no binutils needed on the box)."""
    import struct
    b = open(path, "rb").read()
    assert b[:4] == b"\x7fELF" and b[4] == 2 and b[5] == 1          # ELF64, little endian
    shoff, = struct.unpack_from("<Q", b, 0x28)
    shentsize, shnum = struct.unpack_from("<HH", b, 0x3A)
    secs = [struct.unpack_from("<IIQQQQIIQQ", b, shoff + i * shentsize) for i in range(shnum)]
    out = set()
    for (_n, typ, _f, _a, off, size, link, _i, _al, entsize) in secs:
        if typ != 11:                                               # SHT_DYNSYM
            continue
        stroff = secs[link][4]
        for k in range(1, size // entsize):
            name, info, _other, shndx, _val, _sz = struct.unpack_from("<IBBHQQ", b, off + k * entsize)
            if shndx != 0 and (info >> 4) in (1, 2):                # defined, GLOBAL or WEAK
                end = b.index(b"\0", stroff + name)
                out.add(b[stroff + name:end].decode())
    return out


def test_library_exports_every_declared_symbol():
    names = _header_functions()
    assert len(names) >= 20 and "vasr_transcribe_greedy_f32" in names
    lib = C.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/vasr.h but not exported"
    assert set(names) == set(_lib.SIGNATURES), set(names) ^ set(_lib.SIGNATURES)
    assert b"gfx950" in _lib.lib().vasr_version()
    # the measurement / development entry points live in a second header and a second library: the product library
    # exports none of them (and reads no kernel-selection switch from the environment), the devtools build all of both
    dev_names = _header_functions("vasr_devtools.h")
    assert set(dev_names) == set(_lib.DEV_SIGNATURES) and len(dev_names) >= 10, set(dev_names) ^ set(_lib.DEV_SIGNATURES)
    assert not [n for n in dev_names if hasattr(lib, n)], "devtools symbols in the product library"
    dev = C.CDLL(os.path.join(os.path.dirname(_lib.LIB_PATH), "libvasr_hip_dev.so"))
    for n in names + dev_names:
        assert hasattr(dev, n), f"{n} missing from libvasr_hip_dev.so"
    # ... and NOTHING else: the libraries are linked with -fvisibility=hidden + a version script (csrc/vasr.map), so the
    # dynamic symbol table is exactly the C ABI -- no mangled vasr::launch_* internals, no devtools probe
    assert _dynamic_symbols(_lib.LIB_PATH) == set(names)
    assert _dynamic_symbols(os.path.join(os.path.dirname(_lib.LIB_PATH), "libvasr_hip_dev.so")) == set(names + dev_names)
    assert _lib.lib().vasr_abi_version() == _lib.ABI_VERSION == int(re.search(r"#define VASR_ABI_VERSION (\d+)", open(
        os.path.join(ROOT, "include", "vasr.h")).read()).group(1))
    blob = open(_lib.LIB_PATH, "rb").read()
    for switch in (b"VASR_DW_PAIR", b"VASR_PW3_TILE", b"VASR_FUSED", b"VASR_NO_FUSED_RESIDUAL", b"VASR_DEBUG_NO_EPILOGUE"):
        assert switch not in blob, switch
    assert b"VASR_GEMM" in blob and b"VASR_PW3_TILE" in open(os.path.join(os.path.dirname(_lib.LIB_PATH), "libvasr_hip_dev.so"), "rb").read()


def test_create_validates_arguments_without_a_gpu():
    L = _lib.lib()
    h = C.c_void_p()
    assert L.vasr_create(None, C.byref(h)) == -1 and b"null" in L.vasr_last_error()
    with pytest.raises(NotImplementedError):          # n_fft other than 512 -> VASR_ERR_UNSUPPORTED
        _lib.Handle(frontend=dict(sample_rate=16000, n_fft=1024, win_length=320, hop_length=160, n_mels=64,
                                  window=np.ones(320, np.float32), filterbank=np.zeros((64, 513), np.float32)))
    with pytest.raises(ValueError):                   # non-positive block field -> VASR_ERR_INVALID
        _lib.Handle(feat_in=64, blocks=[dict(filters=256, repeat=0, kernel=33, stride=1, dilation=1, residual=0,
                                             separable=1)])
    hd = _lib.Handle(dec_feat_in=1024, num_classes=29)
    assert hd.workspace_bytes(4, mel_frames=100) == 0          # not finalized yet
    if not torch.cuda.is_available():
        with pytest.raises(_lib.VasrError):                     # missing weights / no device: loud failure
            hd.finalize()


def test_pointwise_weight_packing_layout():
    L = _lib.dev_lib()      # include/vasr_devtools.h lives in the devtools build
    cout, cin, m_pad = 29, 64, 128
    w = np.arange(cout * cin, dtype=np.float32).reshape(cout, cin)
    out = np.empty(m_pad * cin, dtype=np.float32)
    _lib.check(L.vasr_pack_pointwise(w.ctypes.data, cout, cin, m_pad, out.ctypes.data), L)
    p = out.reshape(m_pad // 32, cin // 8, 64, 4)
    for mt, g, lane, s in [(0, 0, 0, 0), (0, 3, 37, 2), (0, 7, 63, 3), (3, 1, 5, 1)]:
        m, k = mt * 32 + (lane & 31), g * 8 + 2 * s + (lane >> 5)
        assert p[mt, g, lane, s] == (w[m, k] if m < cout else 0.0)
    with pytest.raises(ValueError):
        _lib.check(L.vasr_pack_pointwise(w.ctypes.data, cout, 63, m_pad, out.ctypes.data), L)


def test_fused_sub_block_form_by_batch_size():
    """Which form a 256-channel sub-block takes (csrc/vasr_internal.h fused_tile_choice, host arithmetic only): the fused
    kernel on 128-frame tiles when those fill the 256 CUs in whole rounds, on 64-frame tiles for the batches in between,
    two kernels where a lone tile's latency would lose -- the sizes DESIGN section 4 measured."""
    L = _lib.dev_lib()
    pick = lambda batch, frames: L.vasr_fused_tile_choice(batch * (int(L.vasr_padded_frames(frames)) // 128), 256)
    assert pick(64, 501) == 128                     # the headline batch: 256 tiles, one round
    assert pick(512, 1501) == 128                   # configs[4]: 6144 tiles, 24 rounds
    assert pick(52, 501) == 128 and pick(48, 501) == 0     # 208 tiles: 81 % of a round; 192 tiles: 75 %, under the 80 % line
    assert pick(32, 501) == 64                      # configs[1]: 128 tiles of 128 frames would fill half the chip
    assert [pick(b, 501) for b in (12, 16, 24)] == [64, 64, 64]
    assert [pick(b, 501) for b in (1, 4, 8, 11)] == [0, 0, 0, 0]          # latency of a lone tile loses to two kernels
    assert [pick(b, 501) for b in (36, 40, 44)] == [0, 0, 0]              # 1.1-1.4 rounds of 64-frame tiles lose as well
    assert pick(64, 516) == 0                       # 64 x 10.3 s = 320 tiles = 1.25 rounds
    assert pick(103, 501) == 128 and pick(80, 501) == 0                   # last round >= 80 % full or not at all
    assert L.vasr_fused_tile_choice(256, 192) == 0 and L.vasr_fused_tile_choice(96, 192) == 64   # CUs held by a running search
    assert L.vasr_fused_tile_choice(0, 256) == 0


def test_beam_search_form_by_batch_size():
    """Which kernel searches a batch (host arithmetic only, vasr_beam_workgroups = compute units the search occupies): up to 64
    utterances four wavefronts and a compute unit per utterance (csrc/beam_group.hip; the reference serves batch 1,
    infer.py:181-192; at BASELINE configs[3] the shorter stay in the way of the next acoustic pass is worth the wider one,
    beam_group_width), beyond that one wavefront per utterance and four utterances per unit (csrc/beam_wave.hip)."""
    L = _lib.lib()
    assert [L.vasr_beam_workgroups(b) for b in (0, 1, 2, 8, 15, 16, 64)] == [0, 1, 2, 8, 15, 16, 64]
    assert [L.vasr_beam_workgroups(b) for b in (65, 66, 128, 512)] == [17, 17, 32, 128]


@pytest.mark.parametrize("K,dil", [(33, 1), (39, 1), (51, 1), (63, 1), (75, 1), (87, 2)])
def test_depthwise_tap_tables_form_the_toeplitz_product(K, dil):
    """vasr_pack_depthwise_taps (host side of encoder_dw_mfma.hip): the [hi | lo] fp16 table of a channel, read the way the
    kernel builds its A fragments -- A[m][k] = table[15 - m + k], 16 output offsets x 32 NS window samples -- times
    windows of 16 consecutive outputs cut from the zero-padded row, B[k][n] = row[16 n - PADL + k], IS the "same"-padded
    depthwise convolution of jasper.py:60-65 / :119-132; hi + lo carries the scaled tap to 2^-22."""
    L = _lib.dev_lib()      # include/vasr_devtools.h lives in the devtools build
    tsz = int(L.vasr_depthwise_mfma_table_size(K, dil))
    pad = (dil * K) // 2 - 1 if dil > 1 else K // 2
    padl = (pad + 3) & ~3
    ns = (15 + dil * (K - 1) + (padl - pad) + 1 + 31) // 32
    assert tsz == 32 * ns + 16
    rng = np.random.default_rng(K)
    C_ = 3
    w = (rng.standard_normal((C_, K)) / np.sqrt(K)).astype(np.float32)
    w[1] *= 1e-4
    tab, inv = np.empty((C_, tsz), dtype=np.uint32), np.empty(C_, dtype=np.float32)
    _lib.check(L.vasr_pack_depthwise_taps(w.ctypes.data, C_, K, dil, tab.ctypes.data, inv.ctypes.data), L)
    hi = (tab & 0xFFFF).astype(np.uint16).view(np.float16).astype(np.float64)
    lo = (tab >> 16).astype(np.uint16).view(np.float16).astype(np.float64)
    T = 64                                                        # four windows of 16 outputs
    x = rng.standard_normal((C_, T))
    row = np.zeros((C_, padl + T + 32 * ns))                      # row[padl + t] = x[t], zeros around it
    row[:, padl:padl + T] = x
    m, k = np.arange(16)[:, None], np.arange(32 * ns)[None, :]
    for c in range(C_):
        assert np.log2(inv[c]) == np.round(np.log2(inv[c]))       # power-of-two scale
        assert 2.0 ** 14 <= np.abs(hi[c]).max() < 2.0 ** 15
        A = (hi[c] + lo[c])[15 - m + k] * float(inv[c])           # [16][32 ns]
        got = np.concatenate([A @ row[c, 16 * n:16 * n + 32 * ns] for n in range(T // 16)])
        xp = np.zeros(T + 2 * pad)
        xp[pad:pad + T] = x[c]
        ref = np.array([sum(float(w[c, j]) * xp[t + dil * j] for j in range(K)) for t in range(T)])
        assert np.abs(got - ref).max() <= 2.0 ** -20 * np.abs(w[c]).max() * np.abs(x[c]).sum()
    assert int(L.vasr_depthwise_mfma_table_size(35, 1)) == 0      # shapes without an instantiation
    with pytest.raises(ValueError):
        _lib.check(L.vasr_pack_depthwise_taps(w.ctypes.data, C_, 35, 1, tab.ctypes.data, inv.ctypes.data), L)


def test_toeplitz_kernel_lane_algebra():
    """The index algebra of dw_toeplitz_kernel (encoder_dw_mfma.hip) replayed lane by lane in numpy, K = 75: staged row
    sample tau <- frame t0 - PADL + tau; B fragment of lane (n = lane & 15, kg = lane >> 4), group q, step s = samples
    256 q + 16 n + 32 s + 8 kg + e; A fragment of lane (m = lane & 15, kg) = table[15 - m + 32 s + 8 kg + e];
    v_mfma_f32_16x16x32 D: lane n + 16 g holds rows 4 g + r of column n; after the rotation lane L (pulling from lane
    (L >> 2) + 16 (L & 3)) holds frames t0 + 256 q + 4 L + r -- which must be the convolution's."""
    L = _lib.dev_lib()      # include/vasr_devtools.h lives in the devtools build
    K, pad = 75, 37
    padl, ns = (pad + 3) & ~3, 3
    tsz = int(L.vasr_depthwise_mfma_table_size(K, 1))
    rng = np.random.default_rng(5)
    w = (rng.standard_normal((1, K)) / np.sqrt(K)).astype(np.float32)
    tab, inv = np.empty((1, tsz), dtype=np.uint32), np.empty(1, dtype=np.float32)
    _lib.check(L.vasr_pack_depthwise_taps(w.ctypes.data, 1, K, 1, tab.ctypes.data, inv.ctypes.data), L)
    taps = ((tab[0] & 0xFFFF).astype(np.uint16).view(np.float16).astype(np.float64) +
            (tab[0] >> 16).astype(np.uint16).view(np.float16).astype(np.float64)) * float(inv[0])
    T, t0 = 1200, 512                                              # second 512-frame tile of a 1200-frame row
    x = rng.standard_normal(T)
    rows = 512 - 16 + 32 * ns
    staged = np.array([x[t] if 0 <= (t := t0 - padl + tau) < T else 0.0 for tau in range(rows)])
    lanes = np.arange(64)
    n16, kg = lanes & 15, lanes >> 4
    out = np.zeros(512)
    for q in range(2):
        d = np.zeros((64, 4))                                      # accumulators: lane, register
        for s in range(ns):
            a = np.stack([taps[15 - n16 + 32 * s + 8 * kg + e] for e in range(8)], 1)             # [lane][e]
            b = np.stack([staged[256 * q + 16 * n16 + 32 * s + 8 * kg + e] for e in range(8)], 1)
            A = np.zeros((16, 32)); B = np.zeros((32, 16))
            for l in range(64):
                A[n16[l], 8 * kg[l]:8 * kg[l] + 8] = a[l]
                B[8 * kg[l]:8 * kg[l] + 8, n16[l]] = b[l]
            D = A @ B
            for l in range(64):
                d[l] += D[4 * (l >> 4):4 * (l >> 4) + 4, l & 15]
        pulled = d[(lanes >> 2) + 16 * (lanes & 3)]
        for l in range(64):
            out[256 * q + 4 * l:256 * q + 4 * l + 4] = pulled[l]
    xp = np.concatenate([np.zeros(pad), x, np.zeros(pad + 512)])
    ref = np.array([np.dot(w[0].astype(np.float64), xp[t0 + t:t0 + t + K]) for t in range(512)])
    assert np.abs(out - ref).max() <= 1e-5 * np.abs(ref).max()


def test_builtin_configs_and_state_dict_layout():
    for name, n_blocks, n_labels, n_keys in (("quartznet12x1_vi", 15, 90, 182), ("quartznet15x5", 18, 28, 635)):
        cfg = configs.builtin(name)
        jas = cfg["JasperEncoder"]["jasper"]
        assert len(jas) == n_blocks and len(cfg["labels"]) == n_labels
        NeuralModuleFactory(placement=DeviceType.CPU)
        enc = nemo_asr.JasperEncoder(feat_in=64, **cfg["JasperEncoder"])
        sd = synth.encoder_state_dict(jas, 64, 0)
        assert len(enc.state_dict()) == n_keys                      # 182 / 635 keys
        assert set(enc.state_dict()) == set(sd)
        for k, v in enc.state_dict().items():
            assert tuple(v.shape) == tuple(np.shape(sd[k])), k
        enc.load_state_dict({k: torch.as_tensor(v) for k, v in sd.items()})      # strict, like restore_from
    with pytest.raises(ValueError):
        configs.builtin("quartznet99")
    legacy = configs.normalize_definition({"sample_rate": 16000, "AudioPreprocessing": {"feat_type": "logfbank", "n_fft": 512}})
    assert "feat_type" not in legacy["AudioToMelSpectrogramPreprocessor"]


def _wire(placement=DeviceType.CPU):
    cfg = configs.builtin("quartznet12x1_vi")
    nf = NeuralModuleFactory(placement=placement)
    dl = nemo_asr.AudioDataLayer(sample_rate=16000)
    pre = nemo_asr.AudioToMelSpectrogramPreprocessor(**dict(cfg["AudioToMelSpectrogramPreprocessor"], dither=0, pad_to=0))
    enc = nemo_asr.JasperEncoder(feat_in=64, **cfg["JasperEncoder"])
    dec = nemo_asr.JasperDecoderForCTC(feat_in=1024, num_classes=len(cfg["labels"]))
    greedy = nemo_asr.GreedyCTCDecoder()
    return cfg, nf, dl, pre, enc, dec, greedy


def test_dag_wiring_ports_and_type_checks():
    cfg, nf, dl, pre, enc, dec, greedy = _wire()
    sig, sig_len = dl()
    assert isinstance(sig, NmTensor) and sig.name == "audio_signal"
    mel, mel_len = pre(input_signal=sig, length=sig_len)
    encoded, enc_len = enc(audio_signal=mel, length=mel_len)        # MelSpectrogramType is-a SpectrogramType
    logp = dec(encoder_output=encoded)
    pred = greedy(log_probs=logp)
    assert list(pre.output_ports) == ["processed_signal", "processed_length"]
    assert list(enc.output_ports) == ["outputs", "encoded_lengths"]
    assert str(enc) == "JasperEncoder" and str(dec) == "JasperDecoderForCTC"       # checkpoint file stems
    with pytest.raises(NeuralPortNameMismatchError):
        enc(audio=mel, length=mel_len)
    with pytest.raises(NeuralPortNmTensorMismatchError):
        dec(encoder_output=mel_len)                                  # lengths into an acoustic port
    with pytest.raises(NeuralPortNmTensorMismatchError):
        greedy(log_probs=encoded)                                    # ('B','D','T') into ('B','T','D')
    chain = nf._trainer.topo_sort([pred])
    assert [type(e[0]).__name__ for e in chain] == ["AudioDataLayer", "AudioToMelSpectrogramPreprocessor",
                                                    "JasperEncoder", "JasperDecoderForCTC", "GreedyCTCDecoder"]


def test_constructor_errors_mirror_the_reference():
    NeuralModuleFactory(placement=DeviceType.CPU)
    with pytest.raises(ValueError):     # audio_preprocessing.py:339-344
        nemo_asr.AudioToMelSpectrogramPreprocessor(window_size=0.02, n_window_size=320)
    with pytest.raises(ValueError):     # parts/features.py:222-227
        nemo_asr.AudioToMelSpectrogramPreprocessor(log_zero_guard_type="bogus")
    with pytest.raises(ValueError):     # parts/jasper.py:61-62
        nemo_asr.JasperEncoder(jasper=[dict(filters=256, repeat=1, kernel=[33], stride=[2], dilation=[2], dropout=0.0,
                                            residual=False, separable=True)], activation="relu", feat_in=64)
    # the reference's own defaults (dither=1e-5, pad_to=16) and the 15x5 YAML's stft_conv=true must construct;
    # stft_conv selects torch_stft's periodic window
    pre = nemo_asr.AudioToMelSpectrogramPreprocessor()
    assert pre.pad_to == 16 and pre.dither == 1e-5
    sym = nemo_asr.AudioToMelSpectrogramPreprocessor(dither=0)._desc["window"]
    per = nemo_asr.AudioToMelSpectrogramPreprocessor(dither=0, stft_conv=True)._desc["window"]
    from scipy.signal import get_window
    assert np.abs(per - get_window("hann", 320, fftbins=True)).max() < 1e-6
    assert np.abs(sym - get_window("hann", 320, fftbins=False)).max() < 1e-6 and np.abs(sym - per).max() > 1e-3
    assert nemo_asr.AudioToMelSpectrogramPreprocessor(dither=1e-2).dither == 1e-2     # applied in forward (features.py:250-251)
    with pytest.raises(TypeError):      # features.py:209: the reference's own constructor fails on the advertised "max"
        nemo_asr.AudioToMelSpectrogramPreprocessor(dither=0, pad_to="max")
    if not torch.cuda.is_available():
        with pytest.raises(ValueError):     # neural_factory.py:320-330
            NeuralModuleFactory(placement=DeviceType.GPU)


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the loud failure on a box without a HIP device")
def test_no_silent_cpu_fallback():
    cfg, nf, dl, pre, enc, dec, greedy = _wire()
    x = torch.zeros(1, 4000)
    n = torch.tensor([4000])
    for call in (lambda: pre(force_pt=True, input_signal=x, length=n),
                 lambda: enc(force_pt=True, audio_signal=torch.zeros(1, 64, 26), length=torch.tensor([25])),
                 lambda: dec(force_pt=True, encoder_output=torch.zeros(1, 1024, 13)),
                 lambda: greedy(force_pt=True, log_probs=torch.zeros(1, 13, 91))):
        with pytest.raises(_lib.VasrError):
            call()
    from viet_asr_amd.engine import QuartzNetCTC
    with pytest.raises(_lib.VasrError):
        QuartzNetCTC(cfg, {}, {})


def test_synthetic_audio_is_padded_like_the_collate():
    sig, lens = synth.audio_batch(5, 16000, seed=9, ragged=True)
    assert sig.shape == (5, 16000) and lens.max() == 16000 and lens.min() >= 8000
    for b in range(5):
        assert not sig[b, lens[b]:].any()
    dl = nemo_asr.AudioDataLayer(16000) if NeuralModuleFactory(placement=DeviceType.CPU) else None
    dl.set_batch([sig[b, :lens[b]] for b in range(5)])
    a, l = next(iter(dl))
    assert torch.equal(a, torch.from_numpy(sig)) and l.tolist() == lens.tolist()
    with pytest.raises(StopIteration):
        next(dl)


def test_bench_rate_arithmetic_and_core_count_without_a_gpu():
    """bench.py's per-class rates are work-that-ran over time-it-took: the dominant kernel (plain GEMM launches) under
    `roofline`, fused launches on their own, both together as the conservative family figure; cpu_baseline counts the
    CPUs the process can really use (affinity cut down by the cgroup quota), not os.cpu_count()."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("vasr_bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    steps = 10
    prof = {"frontend": dict(ms=1.0, launches=30, flops=0.0, bytes=0.0),
            "depthwise": dict(ms=10.0, launches=470, flops=0.0, bytes=10 * 6.0e9),
            "pointwise": dict(ms=28.0, launches=480, flops=10 * 9.42e11, bytes=0.0),
            "head": dict(ms=0.7, launches=30, flops=0.0, bytes=0.0),
            "fused": dict(ms=9.4, launches=300, flops=10 * 1.51e11, bytes=10 * 2.17e9)}
    cr = bench.class_rates(prof, steps, "f16x2")
    assert abs(cr["exec_tflops"] - 3 * 9.42e11 / 2.8e-3 / 1e12) < 1e-6                   # plain GEMMs only
    assert abs(cr["fam_exec_tflops"] - 3 * (9.42e11 + 1.51e11) / 3.74e-3 / 1e12) < 1e-6   # + fused, whole duration
    assert cr["fam_exec_tflops"] < cr["exec_tflops"] and cr["peak"] == bench.PEAK_16BIT_MFMA_TFLOPS
    assert abs(cr["dw_gbs"] - 6.0e9 / 1e-3 / 1e9) < 1e-6 and abs(cr["fu_gbs"] - 2.17e9 / 0.94e-3 / 1e9) < 1e-6
    assert bench.class_rates(prof, steps, "fp32")["peak"] == bench.PEAK_F32_MFMA_TFLOPS
    n, how = bench.usable_cores()
    assert 1 <= n <= len(os.sched_getaffinity(0)) and "affinity" in how
    assert set(bench.WORKLOADS) == {2, 3, 4, 5} and bench.WORKLOADS[5][1] * 8 == bench.JOB_CLIPS
