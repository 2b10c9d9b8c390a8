import ctypes as C, sys, os, numpy as np, torch
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests", "devtools"))
import viet_asr_amd
from viet_asr_amd import configs, synth, stages, _lib
from viet_asr_amd.engine import QuartzNetCTC
import stress_attack as SA
P = C.CDLL(os.path.join(os.path.dirname(_lib.LIB_PATH), "probe_mfma_attacker.so"))
P.mfma_attacker_launch.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
sink = torch.zeros(4, device="cuda")
cfg = configs.builtin("quartznet12x1_vi"); jas = cfg["JasperEncoder"]["jasper"]
eng = QuartzNetCTC(cfg, synth.encoder_state_dict(jas, 64, 5), synth.decoder_state_dict(1024, 91, 5))
arpa_dir = None
for B, L in ((8, 48000), (64, 160000), (1, 60000)):
    sig, lens = synth.audio_batch(B, L, 3, ragged=True)
    w, n = torch.from_numpy(sig).cuda(), torch.from_numpy(lens).cuda()
    for lds in (0, 8192, 24576):
        att = lambda lds=lds: P.mfma_attacker_launch(4096, lds, 3000, sink.data_ptr(), torch.cuda.current_stream().cuda_stream)
        for name, fn in (("melspec", lambda: stages.melspec(eng.handle, w, n)), ("forward", lambda: eng.forward(w, n, want_logp=True))):
            calls, bad = SA.attack(fn, 2.0, att)
            print(f"{name:8s} {B} x {L} | synthetic f16 MFMA attacker with {lds:5d} B of LDS: calls {calls} wrong {bad}", flush=True)
