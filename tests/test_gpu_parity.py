"""-m gpu: HIP path (through the C ABI) vs the golden fixtures of the real reference and vs the oracle."""
import numpy as np
import pytest
import torch

from conftest import GOLDEN_CASES, load_golden

pytestmark = pytest.mark.gpu

# Tolerances (fp32 path, stated per north_star): the HIP kernels accumulate in a different order
# than MKL-DNN / pocketfft, so values agree to fp32 round-off amplified through <= 86 layers;
# predictions and transcripts must be IDENTICAL (fixture margins are >= 7e-2, five orders above).
MEL_TOL = 2e-4
LOGP_TOL = 2e-3


def _engine(cfg, enc_sd, dec_sd):
    from viet_asr_amd.engine import QuartzNetCTC
    return QuartzNetCTC(cfg, enc_sd, dec_sd)


@pytest.mark.parametrize("name", GOLDEN_CASES)
def test_fused_path_matches_reference_goldens(gpu, name):
    g, cfg, sig, lens, enc_sd, dec_sd = load_golden(name)
    eng = _engine(cfg, enc_sd, dec_sd)
    r = eng.forward(torch.from_numpy(sig).to(gpu), torch.from_numpy(lens).to(gpu), want_logp=True)
    torch.cuda.synchronize()
    logp = r["logp"].cpu().numpy()
    assert logp.shape == g["logp"].shape
    assert np.isfinite(logp).all()
    err = np.abs(logp - g["logp"]).max()
    assert err <= LOGP_TOL, err
    assert (r["enc_len"].cpu().numpy() == g["enc_len"]).all()
    assert (r["pred"].cpu().numpy() == g["pred"]).all()
    assert eng.texts(r["ids"], r["id_len"]) == [str(s) for s in g["hyp"]]
