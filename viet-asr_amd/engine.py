"""QuartzNetCTC: the fused wav -> transcript fast path (one C-ABI call per batch).

This is what bench.py times and what ``VietASR.transcribe_batch`` uses.  It exists *in addition
to* the per-module NeuralModule classes in asr.py (same kernels, port tensors materialised);
both sit on libvasr_hip.so.  PyTorch is used for device memory and streams only.
"""
import numpy as np
import torch

from . import _lib
from .frontend_tables import frontend_description


def _stream_ptr():
    return torch.cuda.current_stream().cuda_stream


def _require_gpu():
    if not torch.cuda.is_available():
        raise _lib.VasrError("viet-asr_amd needs a HIP device (torch.cuda.is_available() is False); "
                             "there is no CPU fallback for this path")


def blocks_from_config(jasper_cfg):
    """YAML block list (configs/*.yaml JasperEncoder.jasper) -> vasr_block_desc dicts."""
    def one(v):
        return int(v[0] if isinstance(v, (list, tuple)) else v)
    out = []
    for l in jasper_cfg:
        for unsupported in ("residual_dense", "se"):
            if l.get(unsupported, False):
                raise NotImplementedError(f"JasperBlock option {unsupported!r} is not implemented")
        if l.get("groups", 1) != 1 or l.get("heads", -1) != -1 or float(l.get("kernel_size_factor", 1.0)) != 1.0:
            raise NotImplementedError("groups/heads/kernel_size_factor other than the defaults are not implemented")
        out.append(dict(filters=int(l["filters"]), repeat=int(l["repeat"]), kernel=one(l["kernel"]),
                        stride=one(l["stride"]), dilation=one(l["dilation"]),
                        residual=int(bool(l["residual"])), separable=int(bool(l.get("separable", False)))))
    return out


class QuartzNetCTC:
    def __init__(self, model_definition, encoder_state, decoder_state, device="cuda:0", gemm=None):
        """gemm: None (library default: "bf16x3"), "bf16x3", "fp32" or the reduced-precision opt-in "bf16x2" --
        see vasr_set_gemm_mode in include/vasr.h."""
        _require_gpu()
        self.device = torch.device(device)
        self.labels = list(model_definition["labels"])
        pre = dict(model_definition["AudioToMelSpectrogramPreprocessor"])
        jas = model_definition["JasperEncoder"]["jasper"]
        enc_cfg = model_definition["JasperEncoder"]
        if enc_cfg.get("activation", "relu") != "relu" or not enc_cfg.get("conv_mask", True):
            raise NotImplementedError("only activation='relu', conv_mask=True is implemented")
        self.frontend = frontend_description(pre)
        self.hop = self.frontend["hop_length"]
        with torch.cuda.device(self.device):
            self.handle = _lib.Handle(frontend=self.frontend, feat_in=pre.get("features", 64),
                                      blocks=blocks_from_config(jas), dec_feat_in=jas[-1]["filters"],
                                      num_classes=len(self.labels) + 1)
            self.handle.load_state_dict(encoder_state)
            self.handle.load_state_dict(decoder_state)
            self.handle.finalize()
            if gemm is not None:
                self.handle.set_gemm_mode(gemm)
        self._ws = None

    # -- shapes
    def frames(self, samples):
        t = self.handle.mel_frames(samples)
        return t, self.handle.encoded_frames(t)

    def _workspace(self, batch, samples):
        need = self.handle.workspace_bytes(batch, samples=samples)
        if self._ws is None or self._ws.numel() < need:
            self._ws = None
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._ws

    def forward(self, wav, length, want_logp=False, want_pred=True):
        """wav [B, L] f32 cuda (rows zero padded), length [B] i64 cuda.

        Returns dict(ids [B,T'] i32, id_len [B] i32, pred [B,T'] i64, enc_len [B] f32, logp or None).
        Everything is enqueued on the current stream; nothing synchronises.
        """
        if wav.device.type != "cuda" or wav.dtype != torch.float32 or not wav.is_contiguous():
            raise ValueError("wav must be a contiguous float32 cuda tensor")
        if length.dtype != torch.int64 or length.device != wav.device:
            raise ValueError("length must be an int64 tensor on the same device")
        B, L = wav.shape
        _, t1 = self.frames(L)
        ws = self._workspace(B, L)
        dev = wav.device
        ids = torch.empty((B, t1), dtype=torch.int32, device=dev)
        id_len = torch.empty((B,), dtype=torch.int32, device=dev)
        pred = torch.empty((B, t1), dtype=torch.int64, device=dev) if want_pred else None
        enc_len = torch.empty((B,), dtype=torch.float32, device=dev)
        logp = torch.empty((B, t1, len(self.labels) + 1), dtype=torch.float32, device=dev) if want_logp else None
        _lib.check(_lib.lib().vasr_transcribe_greedy_f32(
            self.handle.h, wav.data_ptr(), length.data_ptr(), B, L,
            pred.data_ptr() if pred is not None else None, ids.data_ptr(), id_len.data_ptr(),
            logp.data_ptr() if logp is not None else None, enc_len.data_ptr(),
            ws.data_ptr(), ws.numel(), _stream_ptr()))
        return dict(ids=ids, id_len=id_len, pred=pred, enc_len=enc_len, logp=logp)

    def texts(self, ids, id_len):
        """Host side of helpers.py:32 -- ''.join(labels[c]) over the collapsed ids."""
        ids = ids.cpu().numpy()
        n = id_len.cpu().numpy()
        return ["".join(self.labels[c] for c in ids[b, : n[b]]) for b in range(ids.shape[0])]

    def transcribe(self, signals):
        """List of 1-D float arrays (16 kHz) -> list of strings; zero-pad-to-max collate
        (parts/dataset.py:14-53)."""
        lens = np.array([len(s) for s in signals], dtype=np.int64)
        L = int(lens.max())
        batch = np.zeros((len(signals), L), dtype=np.float32)
        for i, s in enumerate(signals):
            batch[i, : len(s)] = np.asarray(s, dtype=np.float32)
        wav = torch.from_numpy(batch).to(self.device)
        ln = torch.from_numpy(lens).to(self.device)
        r = self.forward(wav, ln)
        return self.texts(r["ids"], r["id_len"])
