#!/usr/bin/env python3
"""Stage-by-stage GPU-vs-oracle diagnostics (dev tool; run through gpurun)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import viet_asr_amd  # noqa
from viet_asr_amd import _lib, configs, stages, synth
from viet_asr_amd.engine import QuartzNetCTC, blocks_from_config
from viet_asr_amd.frontend_tables import frontend_description
from oracle import quartznet_oracle as O


def stage_check(model, B, L, seed, ragged=True):
    cfg = configs.builtin(model)
    jas = cfg["JasperEncoder"]["jasper"]
    enc_sd = synth.encoder_state_dict(jas, 64, seed)
    dec_sd = synth.decoder_state_dict(1024, len(cfg["labels"]) + 1, seed)
    sig, lens = synth.audio_batch(B, L, seed, ragged)
    ref = O.forward_all(sig, lens, enc_sd, dec_sd, jas)
    dev = torch.device("cuda:0")
    fe = frontend_description(cfg["AudioToMelSpectrogramPreprocessor"])
    hp = _lib.Handle(frontend=fe); hp.finalize()
    he = _lib.Handle(feat_in=64, blocks=blocks_from_config(jas)); he.load_state_dict(enc_sd); he.finalize()
    hd = _lib.Handle(dec_feat_in=1024, num_classes=len(cfg["labels"]) + 1); hd.load_state_dict(dec_sd); hd.finalize()
    wav, ln = torch.from_numpy(sig).to(dev), torch.from_numpy(lens).to(dev)
    mel, seq = stages.melspec(hp, wav, ln)
    torch.cuda.synchronize()
    print(f"[{model} B={B} L={L}] mel err {np.abs(mel.cpu().numpy() - ref['mel'].numpy()).max():.3e}  "
          f"seq ok {(seq.cpu() == ref['seq']).all().item()}")
    # encoder fed with the ORACLE mel, to isolate stages
    enc, elen = stages.encoder(he, ref["mel"].to(dev), ref["seq"].to(dev), 1024)
    torch.cuda.synchronize()
    e = np.abs(enc.cpu().numpy() - ref["enc"].numpy())
    print(f"   enc err max {e.max():.3e} mean {e.mean():.3e} (|ref| max {ref['enc'].abs().max():.3f})  "
          f"enc_len ok {(elen.cpu() == ref['enc_len']).all().item()}")
    logp = stages.decoder(hd, ref["enc"].to(dev))
    torch.cuda.synchronize()
    print(f"   logp err (oracle enc in) {np.abs(logp.cpu().numpy() - ref['logp'].numpy()).max():.3e}")
    pred = stages.greedy_argmax(ref["logp"].to(dev))
    ids, n = stages.ctc_collapse(ref["pred"].to(dev), len(cfg["labels"]))
    torch.cuda.synchronize()
    print(f"   argmax ok {(pred.cpu() == ref['pred']).all().item()}")
    idc, nc = ids.cpu().numpy(), n.cpu().numpy()
    okc = all(list(idc[b, :nc[b]]) == O.ctc_collapse_ids(ref["pred"][b].numpy(), len(cfg["labels"])) for b in range(B))
    print(f"   collapse ok {okc}")
    eng = QuartzNetCTC(cfg, enc_sd, dec_sd)
    r = eng.forward(wav, ln, want_logp=True)
    torch.cuda.synchronize()
    lerr = np.abs(r["logp"].cpu().numpy() - ref["logp"].numpy()).max()
    top2 = torch.topk(ref["logp"], 2, dim=-1).values
    print(f"   fused: logp err {lerr:.3e} pred mismatches {(r['pred'].cpu() != ref['pred']).sum().item()} "
          f"/ {ref['pred'].numel()}  min margin {(top2[..., 0] - top2[..., 1]).min().item():.3e}")


if __name__ == "__main__":
    print(torch.cuda.get_device_name(0))
    stage_check("quartznet12x1_vi", 1, 4000, 4, ragged=False)
    stage_check("quartznet12x1_vi", 3, 32480, 1)
    stage_check("quartznet15x5", 2, 20321, 3)
    stage_check("quartznet12x1_vi", 4, 160000, 7)
