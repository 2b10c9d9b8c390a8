import ctypes as C, sys, os, numpy as np, torch
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests", "devtools"))
import viet_asr_amd
from viet_asr_amd import configs, synth, stages, _lib
from viet_asr_amd.frontend_tables import frontend_description
import stress_attack as SA
P = C.CDLL(os.path.join(R, "viet-asr_amd", "lib", "probe_mfma_attacker.so"))
P.mfma_attacker_launch.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
sink = torch.zeros(1 << 18, device="cuda"); src = torch.randn(1 << 22, device="cuda")
cfg = configs.builtin("quartznet12x1_vi")
h = _lib.Handle(frontend=frontend_description(dict(cfg["AudioToMelSpectrogramPreprocessor"]))); h.finalize()
sig, lens = synth.audio_batch(64, 160000, 3, ragged=True)
w, n = torch.from_numpy(sig).cuda(), torch.from_numpy(lens).cuda()
tag = os.path.basename(_lib.LIB_PATH)
for flags in (0, 1, 2, 3, 7, 15):
    att = lambda f=flags: P.mfma_attacker_launch(2048, 24576, 600, f, sink.data_ptr(), src.data_ptr(), torch.cuda.current_stream().cuda_stream)
    calls, bad = SA.attack(lambda: stages.melspec(h, w, n), 2.0, att)
    print(f"[{tag}] melspec 64 x 160000 | synthetic attacker flags {flags:2d} (1 LDS+barrier, 2 global loads, 4 packed conversions, 8 stores): calls {calls} wrong {bad}", flush=True)
calls, bad = SA.attack(lambda: stages.melspec(h, w, n), 2.0)
print(f"[{tag}] melspec 64 x 160000 | torch fp16 bmm attacker: calls {calls} wrong {bad}")
