"""Greedy CTC post-processing: counterpart of nemo/collections/asr/helpers.py:7-33, 207-208.

The reference collapses predictions in a Python loop over every frame of every utterance; here the
collapse runs on the device (vasr_ctc_collapse: ballot + popcount compaction, one wavefront per
utterance) and only the id -> label join happens on the host.  Like the reference it walks ALL T'
frames, including padded ones (quirk Q4).
"""
import torch

from . import stages


def ctc_decoder_predictions_tensor(tensor, labels):
    """[B, T'] predictions -> list of B strings (helpers.py:7-33)."""
    blank_id = len(labels)
    t = torch.as_tensor(tensor).long()
    if t.device.type != "cuda":
        if not torch.cuda.is_available():
            from ._lib import VasrError
            raise VasrError("viet-asr_amd needs a HIP device; there is no CPU fallback for this path")
        t = t.cuda()
    ids, n = stages.ctc_collapse(t, blank_id)
    ids, n = ids.cpu().numpy(), n.cpu().numpy()
    return ["".join(labels[c] for c in ids[b, : n[b]]) for b in range(ids.shape[0])]


def post_process_predictions(predictions, labels):
    """helpers.py:207-208 -> __gather_predictions (:120-125): list of [B,T'] tensors -> flat list of strings."""
    results = []
    for prediction in predictions:
        results += ctc_decoder_predictions_tensor(prediction, labels=labels)
    return results
