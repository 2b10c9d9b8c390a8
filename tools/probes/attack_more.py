import sys, os, numpy as np, torch
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests", "devtools"))
import viet_asr_amd
from viet_asr_amd import configs, synth
from viet_asr_amd.engine import QuartzNetCTC
import stress_attack as SA
tot = 0
for model in ("quartznet15x5", "quartznet12x1_vi"):
    cfg = configs.builtin(model); jas = cfg["JasperEncoder"]["jasper"]
    eng = QuartzNetCTC(cfg, synth.encoder_state_dict(jas, 64, 5), synth.decoder_state_dict(1024, len(cfg["labels"]) + 1, 5))
    for B, L in ((2, 64000), (4, 32000), (5, 160000), (8, 48000), (12, 160000), (16, 80000), (24, 160000), (32, 160000), (48, 64000), (64, 160000)):
        sig, lens = synth.audio_batch(B, L, B, ragged=True)
        for pcm in (False, True):
            w = torch.from_numpy(np.round(sig * 20000).astype(np.int16) if pcm else sig).cuda(); n = torch.from_numpy(lens).cuda()
            for att in ("bmm16", "matmul_bf16"):
                calls, bad = SA.attack(lambda: eng.forward(w, n, want_logp=True), 1.0, SA.attackers()[att])
                tot += bad
                if bad: print(f"{model} {B} x {L} pcm {pcm} attacker {att}: calls {calls} wrong {bad}", flush=True)
print("TOTAL wrong", tot)
