"""Dumps the inputs of the Python-side victim (tests/devtools/stress_attack.py, tools/probes/conc_probe13.py) for the stand-alone reproducer."""
import os, sys
import numpy as np
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo"); sys.path.insert(0, R)
import viet_asr_amd
from viet_asr_amd import configs, synth
from viet_asr_amd.frontend_tables import frontend_description
d = sys.argv[1]; os.makedirs(d, exist_ok=True)
fe = frontend_description(dict(configs.builtin("quartznet12x1_vi")["AudioToMelSpectrogramPreprocessor"], normalize=None))
sig, lens = synth.audio_batch(64, 160000, 3, ragged=False)
sig.astype(np.float32).tofile(os.path.join(d, "wav.bin")); np.asarray(fe["window"], dtype=np.float32).tofile(os.path.join(d, "win.bin"))
np.asarray(fe["filterbank"], dtype=np.float32).tofile(os.path.join(d, "fb.bin"))
print("dumped", sig.shape, np.asarray(fe["window"]).shape, np.asarray(fe["filterbank"]).shape)
