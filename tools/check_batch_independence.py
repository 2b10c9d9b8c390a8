import sys, numpy as np, torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import viet_asr_amd
from viet_asr_amd import configs, synth
from viet_asr_amd.engine import QuartzNetCTC
cfg = configs.builtin("quartznet15x5"); jas = cfg["JasperEncoder"]["jasper"]
eng = QuartzNetCTC(cfg, synth.encoder_state_dict(jas, 64, 3), synth.decoder_state_dict(1024, 29, 3))
for B, L in ((129, 48000), (513, 32000)):
    sig, lens = synth.audio_batch(B, L, 11)
    r = eng.forward(torch.from_numpy(sig).cuda(), torch.from_numpy(lens).cuda(), want_logp=True)
    torch.cuda.synchronize()
    for b in (0, B // 2, B - 1):
        r1 = eng.forward(torch.from_numpy(sig[b:b+1]).cuda(), torch.from_numpy(lens[b:b+1]).cuda(), want_logp=True)
        d = (r["logp"][b] - r1["logp"][0]).abs().max().item()
        same = torch.equal(r["pred"][b], r1["pred"][0])
        print(f"B={B} row {b}: max |logp diff| vs alone {d:.3e}, same pred {same}")
