"""vasr_set_slices: a batch cut into 2-4 parts on internal streams (one part's front end runs next to another part's GEMMs: the
co-residency of profiles/r06_concurrency.txt INSIDE one call).  Row-independent mode promises results that do not depend on it."""
import sys, os, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import viet_asr_amd
from viet_asr_amd import configs, synth
from viet_asr_amd.engine import QuartzNetCTC
for model in ("quartznet12x1_vi", "quartznet15x5"):
    cfg = configs.builtin(model); jas = cfg["JasperEncoder"]["jasper"]
    eng = QuartzNetCTC(cfg, synth.encoder_state_dict(jas, 64, 5), synth.decoder_state_dict(1024, len(cfg["labels"]) + 1, 5))
    for B, L in ((24, 30000), (64, 16000), (40, 9000)):
        sig, lens = synth.audio_batch(B, L, 7 + B, ragged=True)
        w, n = torch.from_numpy(sig).cuda(), torch.from_numpy(lens).cuda()
        eng.handle.set_slices(1)
        want = eng.forward(w, n, want_logp=True, row_independent=True)["logp"].clone(); torch.cuda.synchronize()
        for k in (2, 3, 4):
            eng.handle.set_slices(k)
            bad = 0
            first = None
            for it in range(300):
                r = eng.forward(w, n, want_logp=True, row_independent=True)["logp"]
                torch.cuda.synchronize()
                f = eng.frames(int(lens.max()))[1]
                ok = all(torch.equal(r[b, : eng.frames(int(lens[b]))[1]], want[b, : eng.frames(int(lens[b]))[1]]) for b in range(B))
                bad += int(not ok)
            print(f"{model} {B} x {L} slices {k}: 300 calls, {bad} differ from the unsliced row-independent result", flush=True)
        eng.handle.set_slices(1)
