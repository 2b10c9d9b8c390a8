import sys, os, threading, time, copy, numpy as np, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import viet_asr_amd
from viet_asr_amd import configs, synth, stages, _lib
from viet_asr_amd.engine import QuartzNetCTC
def blk(filters, kernel, repeat, stride=1, residual=False, separable=True, dilation=1):
    return dict(filters=filters, repeat=repeat, kernel=[kernel], stride=[stride], dilation=[dilation], dropout=0.0, residual=residual, separable=separable)
ARCHS = {
  "12x1_vi": None,
  "prologue only (stride-2 sep K33 64->256) + 1x1": [blk(256, 33, 1, stride=2), blk(256, 1, 1, separable=False)],
  "prologue + one 256 K33 sub-block": [blk(256, 33, 1, stride=2), blk(256, 33, 1), blk(256, 1, 1, separable=False)],
  "prologue + 512 K51 x2 residual": [blk(256, 33, 1, stride=2), blk(512, 51, 2, residual=True), blk(256, 1, 1, separable=False)],
  "dense 1x1 only": [blk(256, 1, 1, separable=False)],
}
def engine_for(name, gemm=None):
    import copy as cp
    cfg = cp.deepcopy(configs.builtin("quartznet12x1_vi"))
    if ARCHS[name] is not None:
        cfg["JasperEncoder"]["jasper"] = ARCHS[name]
    jas = cfg["JasperEncoder"]["jasper"]
    return QuartzNetCTC(cfg, synth.encoder_state_dict(jas, 64, 5), synth.decoder_state_dict(jas[-1]["filters"], 91, 5), gemm=gemm)
def trial(name, gemm, shapes):
    eng, eng2 = engine_for(name, gemm), engine_for(name, gemm)
    pool = []
    for i, (B, L) in enumerate(shapes):
        sig, lens = synth.audio_batch(B, L, 50 + i, ragged=True)
        pool.append((torch.from_numpy(sig).cuda(), torch.from_numpy(lens).cuda()))
    want = [eng.forward(w, n, want_logp=True)["logp"].clone() for w, n in pool]; torch.cuda.synchronize()
    stop = [False]; bad = [0]; calls = [0]
    def a():
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            while not stop[0]:
                for i, (w, n) in enumerate(pool):
                    r = eng.forward(w, n, want_logp=True); st.synchronize(); calls[0] += 1
                    if not torch.equal(r["logp"], want[i]): bad[0] += 1
    def b():
        st = torch.cuda.Stream(); i = 0
        with torch.cuda.stream(st):
            while not stop[0]:
                w, n = pool[i % len(pool)]; i += 1
                eng2.forward(w, n, want_logp=True); st.synchronize()
    ta, tb = threading.Thread(target=a), threading.Thread(target=b)
    ta.start(); tb.start(); time.sleep(8); stop[0] = True; ta.join(); tb.join()
    print(f"{name:48s} gemm {gemm or 'f16x2':7s} shapes {shapes}: calls {calls[0]} wrong {bad[0]}", flush=True)
small, mid, big = [(1, 30000), (4, 20000)], [(14, 12000)], [(40, 9000)]
for name in ARCHS:
    trial(name, None, small + mid + big)
for g in ("fp32", "bf16x3"):
    trial("12x1_vi", g, small + mid + big)
for shp in (small, mid, big):
    trial("12x1_vi", None, shp)
