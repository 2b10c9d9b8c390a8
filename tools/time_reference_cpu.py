#!/usr/bin/env python3
"""Dev container only (needs /root/reference): time the IMPORTED reference modules on the CPU next to this repository's
oracle on the same inputs and write profiles/rNN_reference_cpu_devcontainer.json.

    python tools/time_reference_cpu.py profiles/r04_reference_cpu_devcontainer.json

The reference never travels to the GPU box; bench.py's `cpu_baseline` there times the oracle (kind "port").  This file is
the evidence that the port runs at the reference's speed: both call the same ATen CPU ops."""
import json
import os
import sys
import time

import numpy as np
import torch
import yaml

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests", "golden"))
import make_golden as MG  # noqa: E402


def best_of(fn, n=3):
    fn()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return min(ts), float(np.median(ts))


def main():
    out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(REPO, "profiles", "reference_cpu_devcontainer.json")
    sys.path.insert(0, MG.REF)
    MG.install_shims()
    pkg = MG._load_pkg()
    synth = pkg.synth
    from oracle import quartznet_oracle as O
    threads = len(os.sched_getaffinity(0))
    torch.set_num_threads(threads)
    rows = []
    for cfg_file, model, batch, seconds in (("quartznet12x1_vi.yaml", "quartznet12x1_vi", 1, 6.6),
                                            ("quartznet12x1_vi.yaml", "quartznet12x1_vi", 32, 10.0),
                                            ("quartznet15x5.yaml", "quartznet15x5", 4, 10.0),
                                            ("quartznet15x5.yaml", "quartznet15x5", 8, 10.0)):
        cfg = yaml.safe_load(open(os.path.join(MG.REF, "configs", cfg_file), encoding="utf-8"))
        labels = cfg["labels"]
        nf, pre, enc, dec, greedy = MG.build_reference(cfg, labels)
        jas = cfg["JasperEncoder"]["jasper"]
        enc_sd = synth.encoder_state_dict(jas, 64, 3)
        dec_sd = synth.decoder_state_dict(jas[-1]["filters"], len(labels) + 1, 3)
        enc.load_state_dict({k: torch.as_tensor(v) for k, v in enc_sd.items()})
        dec.load_state_dict({k: torch.as_tensor(v) for k, v in dec_sd.items()})
        enc.eval(); dec.eval(); greedy.eval()
        sig, lens = synth.audio_batch(batch, int(seconds * 16000), 3, ragged=False)
        from nemo.collections.asr.helpers import post_process_predictions

        def ref_once():
            with torch.no_grad():
                mel, seq = pre(force_pt=True, input_signal=torch.as_tensor(sig), length=torch.as_tensor(lens))
                e, _ = enc(force_pt=True, audio_signal=mel, length=seq)
                pred = greedy(force_pt=True, log_probs=dec(force_pt=True, encoder_output=e))
            return post_process_predictions([pred], labels)

        def port_once():
            with torch.no_grad():
                return O.ctc_decode_strings(O.forward_all(sig, lens, enc_sd, dec_sd, jas)["pred"], labels)

        assert ref_once() == port_once()
        rb, rm = best_of(ref_once)
        pb, pm = best_of(port_once)
        audio = batch * seconds
        rows.append({"model": model, "batch": batch, "clip_seconds": seconds,
                     "reference_modules": {"best_s": round(rb, 4), "median_s": round(rm, 4), "rtf": round(audio / rb, 1)},
                     "oracle_port": {"best_s": round(pb, 4), "median_s": round(pm, 4), "rtf": round(audio / pb, 1)},
                     "port_over_reference": round(pb / rb, 3), "transcripts_identical": True})
        print(rows[-1])
    out = {"what": "imported reference modules (AudioToMelSpectrogramPreprocessor -> JasperEncoder -> JasperDecoderForCTC -> "
                   "GreedyCTCDecoder -> post_process_predictions, force_pt call convention of actions.py:419-428) vs "
                   "oracle/quartznet_oracle.py on the same seeded inputs and weights; best of 3 after one warm-up",
           "where": "development container (no GPU)", "threads": threads, "torch": torch.__version__, "rows": rows}
    json.dump(out, open(out_path, "w"), indent=1)
    print("wrote", out_path)


if __name__ == "__main__":
    main()
