import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import viet_asr_amd  # noqa: E402,F401  (root shim -> viet-asr_amd/)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")
GOLDEN_CASES = ["vi12x1_b1_tiny", "vi12x1_b3_ragged", "vi12x1_b2_q2_realdec", "en15x5_b2_ragged", "en12x1_b4_hopmult",
                "en15x5_b1_10s"]


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def load_golden(name):
    """-> (golden npz, model definition, audio, lens, encoder sd, decoder sd); inputs regenerated from seeds."""
    from viet_asr_amd import configs, synth
    g = np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)
    cfg = configs.builtin(str(g["cfg_file"]))
    jas = cfg["JasperEncoder"]["jasper"]
    seed = int(g["seed"])
    sig, lens = synth.audio_batch(int(g["batch"]), int(g["samples"]), seed, bool(g["ragged"]))
    enc_sd = synth.encoder_state_dict(jas, 64, seed)
    if str(g["real_decoder"]):
        dec_sd = {"decoder_layers.0.weight": g["dec_weight"], "decoder_layers.0.bias": g["dec_bias"]}
    else:
        dec_sd = synth.decoder_state_dict(jas[-1]["filters"], len(cfg["labels"]) + 1, seed)
    return g, cfg, sig, lens, enc_sd, dec_sd


REAL_AUDIO_CASES = ["real16k_thoisu_5", "real8k_external_2"]


def load_real_audio_golden(name):
    """-> (golden npz, model definition, int16 PCM, sample rate, synthetic encoder sd, REAL Vietnamese head sd, seeded head sd).
    A recording of the reference's audio_samples/ with the outputs of the imported reference (make_golden.py)."""
    from viet_asr_amd import configs, synth
    g = np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)
    head = np.load(os.path.join(GOLDEN_DIR, "vi12x1_b2_q2_realdec.npz"), allow_pickle=False)   # the same shipped checkpoint
    cfg = configs.builtin(str(g["cfg_file"]))
    jas = cfg["JasperEncoder"]["jasper"]
    seed = int(g["seed"])
    real = {"decoder_layers.0.weight": head["dec_weight"], "decoder_layers.0.bias": head["dec_bias"]}
    return (g, cfg, g["pcm"], int(g["sample_rate"]), synth.encoder_state_dict(jas, 64, seed), real,
            synth.decoder_state_dict(1024, len(cfg["labels"]) + 1, seed))


@pytest.fixture(scope="session")
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no HIP device")
    return torch.device("cuda:0")
