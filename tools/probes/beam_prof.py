#!/usr/bin/env python3
"""Section cycle counters of the beam-search kernels at the reference's serving shape (batch 1, QuartzNet12x1 Vietnamese head,
6.6 s, 3-gram LM): ONE launch per case with a -DVASR_BEAM_PROF build (VASR_LIB_PATH), on the model's own posteriors and on
CTC-like ones.  VASR_BEAM_GROUP picks the kernel form (devtools build)."""
import os, sys, tempfile
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import viet_asr_amd  # noqa: F401
from viet_asr_amd import configs, synth
from viet_asr_amd.beam import BeamSearchDecoder, read_arpa
from viet_asr_amd.engine import QuartzNetCTC

dev = torch.device("cuda:0")
cfg = configs.builtin("quartznet12x1_vi")
jas = cfg["JasperEncoder"]["jasper"]
eng = QuartzNetCTC(cfg, synth.encoder_state_dict(jas, 64, 3), synth.decoder_state_dict(jas[-1]["filters"], len(cfg["labels"]) + 1, 3))
sig, lens = synth.audio_batch(1, int(6.6 * 16000), 5)
lp = eng.forward(torch.from_numpy(sig).to(dev), torch.from_numpy(lens).to(dev), want_logp=True, want_pred=False)["logp"]
arpa = os.path.join(tempfile.mkdtemp(prefix="vasr_lm_"), "synthetic3.arpa")
synth.synthetic_arpa(arpa, cfg["labels"], seed=3)
words = sorted(w[0] for w in read_arpa(arpa)[1] if len(w) == 1 and not w[0].startswith("<"))
lp_ctc = torch.from_numpy(synth.ctc_like_log_probs(1, lp.shape[1], cfg["labels"], words, seed=5)).to(dev)
dec = BeamSearchDecoder(cfg["labels"], lm_path=arpa, alpha=0.5, beta=1.5)
for name, x in (("model", lp), ("ctc-like", lp_ctc)):
    for width in [int(w) for w in os.environ.get("WIDTHS", "50,100").split(",")]:
        print(f"== {name} posteriors, beam {width}, {x.shape[1]} frames", flush=True)
        dec.decode_ids(x, width)
        torch.cuda.synchronize()
