"""QuartzNetCTC: the fused wav -> transcript fast path (one C-ABI call per batch).

This is what bench.py times and what ``VietASR.transcribe_batch`` uses.  It exists *in addition
to* the per-module NeuralModule classes in asr.py (same kernels, port tensors materialised);
both sit on libvasr_hip.so.  PyTorch is used for device memory and streams only.
"""
import threading

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


def _pcm_to_float(s):
    """Integer PCM -> float32 scaled by 2^-(bits - 1) (AudioSegment._convert_samples_to_float32, parts/segment.py:61-74; exact for
    int16); anything else unchanged."""
    dt = getattr(s, "dtype", None)
    if dt is not None and dt.kind == "i":
        return s.astype(np.float32) * np.float32(1.0 / 2 ** (8 * dt.itemsize - 1))
    return s


class QuartzNetCTC:
    def __init__(self, model_definition, encoder_state, decoder_state, device="cuda:0", gemm=None):
        """gemm: None (library default: "f16x2"), "f16x2", "bf16x3", "fp32" or the reduced-precision opt-in "bf16x2" --
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
        self._blocks = blocks_from_config(jas)
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
        self._slots, self._copy_stream, self._launched = None, None, 0
        self._row_independent = False
        self._beam_stream = None
        self._beam_inflight = None                         # (done event, workgroups) of the last overlapped search

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

    def forward(self, wav, length, want_logp=False, want_pred=True, row_independent=False):
        """wav [B, L] f32 cuda (rows zero padded) -- or int16 PCM as it sits in a wav file: the front end scales it by 2^-15 as
        it reads it (vasr_transcribe_greedy_pcm16; the same bits as converting first) --, length [B] i64 cuda.

        Returns dict(ids [B,T'] i32, id_len [B] i32, pred [B,T'] i64, enc_len [B] f32, logp or None).
        Everything is enqueued on the current stream; nothing synchronises.

        row_independent=False is the reference's batched semantics (a row's result depends on the padded batch: reflect
        padding at the padded end, padded frames decoded).  True makes ids / id_len of every row what a batch-1 call on
        that row alone returns, bit for bit (vasr_set_row_independent in include/vasr.h); every length must then
        exceed n_fft / 2, which the caller checks (the lengths live on the device here).
        """
        if wav.device.type != "cuda" or wav.dtype not in (torch.float32, torch.int16) or not wav.is_contiguous():
            raise ValueError("wav must be a contiguous float32 (or int16 PCM) cuda tensor")
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
        if bool(row_independent) != self._row_independent:
            self.handle.set_row_independent(row_independent)
            self._row_independent = bool(row_independent)
        entry = _lib.lib().vasr_transcribe_greedy_pcm16 if wav.dtype == torch.int16 else _lib.lib().vasr_transcribe_greedy_f32
        _lib.check(entry(
            self.handle.h, wav.data_ptr(), length.data_ptr(), B, L,
            pred.data_ptr() if pred is not None else None, ids.data_ptr(), id_len.data_ptr(),
            logp.data_ptr() if logp is not None else None, enc_len.data_ptr(),
            ws.data_ptr(), ws.numel(), _stream_ptr()))
        return dict(ids=ids, id_len=id_len, pred=pred, enc_len=enc_len, logp=logp)

    def texts(self, ids, id_len):
        """Host side of helpers.py:32 -- ''.join(labels[c]) over the collapsed ids."""
        ids = ids.cpu().numpy()
        n = id_len.cpu().numpy()
        if (n < 0).any():      # vasr.h: id_len = -1 = the beam search's merge cells overflowed (cannot happen; reported, not hidden)
            raise _lib.VasrError("beam search reported an internal overflow (id_len = -1) for rows %s" % (n < 0).nonzero()[0].tolist())
        return ["".join(self.labels[c] for c in ids[b, : n[b]]) for b in range(ids.shape[0])]

    def transcribe(self, signals, row_independent=False):
        """List of 1-D arrays (model sample rate; float, or int16 PCM) -> list of strings; zero-pad-to-max collate
        (parts/dataset.py:14-53).  row_independent: see forward()."""
        return self.launch(signals, row_independent).texts()

    def transcribe_beam(self, signals, beam_decoder, beam_width, row_independent=True):
        """Batched counterpart of the reference's beam wiring (infer.py:132-139, 159-160; batch 1 there): one forward
        pass for the log-probs, then viet_asr_amd.beam.BeamSearchDecoder over every row.  row_independent=True searches
        each row over its own frames (and reflects it at its own end): the transcripts of batch-1 calls."""
        lens = [len(s) for s in signals]
        if row_independent and min(lens) <= self.frontend["n_fft"] // 2:
            raise ValueError(f"row-independent batching needs more than n_fft/2 = {self.frontend['n_fft'] // 2} samples "
                             f"per signal (got {min(lens)})")
        batch = np.zeros((len(signals), max(lens)), dtype=np.float32)
        for i, s in enumerate(signals):
            batch[i, : lens[i]] = _pcm_to_float(s)
        r = self.forward(torch.from_numpy(batch).to(self.device), torch.tensor(lens, device=self.device),
                         want_logp=True, want_pred=False, row_independent=row_independent)
        own = [self.frames(n)[1] for n in lens] if row_independent else None
        return beam_decoder.decode_batch(r["logp"], beam_width, frames=own)

    def forward_beam(self, wav, length, beam_decoder, beam_width, frames=None, overlap=True):
        """Acoustic pass + beam search (+ LM) of one device-resident batch: the batched form of infer.py:146-160.

        overlap=True runs the search on a side stream: it occupies a compute unit per utterance up to 64 utterances (four
        wavefronts each, 1.3 ms at B = 64 x 501 frames, beam 128), a compute unit per four utterances beyond, so queued behind the
        log-probs of batch k it runs under the kernels of batch k + 1 instead of after them.  Returns dict(ids, id_len,
        score, done): ``done`` is an event on the side stream (None when overlap is off); wait for it -- or
        synchronise the device -- before reading the results from another stream."""
        # a search of the previous batch that is still queued or running holds a compute unit per four utterances while this
        # acoustic pass runs: the GEMM tile choice should fill whole rounds of what is left (vasr_set_busy_cus)
        busy = self._beam_inflight[1] if overlap and self._beam_inflight and not self._beam_inflight[0].query() else 0
        self.handle.set_busy_cus(busy)
        try:
            r = self.forward(wav, length, want_logp=True, want_pred=False)
        finally:
            self.handle.set_busy_cus(0)
        if not overlap:
            ids, n, score = beam_decoder.decode_ids(r["logp"], beam_width, frames)
            return dict(ids=ids, id_len=n, score=score, done=None, enc_len=r["enc_len"])
        if self._beam_stream is None:
            self._beam_stream = torch.cuda.Stream(self.device)
        main = torch.cuda.current_stream(self.device)
        ready = torch.cuda.Event()
        ready.record(main)
        with torch.cuda.stream(self._beam_stream):
            self._beam_stream.wait_event(ready)
            ids, n, score = beam_decoder.decode_ids(r["logp"], beam_width, frames)
            done = torch.cuda.Event()
            done.record(self._beam_stream)
        self._beam_inflight = (done, int(_lib.lib().vasr_beam_workgroups(int(r["logp"].shape[0]))))
        r["logp"].record_stream(self._beam_stream)     # allocated on the main stream, last read on the side stream
        return dict(ids=ids, id_len=n, score=score, done=done, enc_len=r["enc_len"])

    # -- long recordings in bounded memory
    def halo_mel_frames(self):
        """Half-width of the encoder's receptive field in mel frames (even): sum over the depthwise convolutions of
        (K - 1) / 2 * dilation, times the stride product in front of each."""
        h, rate = 0, 1
        for b in self._blocks:
            k = b["kernel"] + (1 if b["kernel"] % 2 == 0 else 0)
            for _ in range(b["repeat"]):
                if b["separable"] or k > 1:
                    h += (k - 1) // 2 * b["dilation"] * rate
                rate *= b["stride"]
                if b["stride"] > 1 and b["repeat"] > 1:
                    raise NotImplementedError("strided block with repeat > 1")
        return h + (h & 1)

    def forward_long(self, wav, chunk_frames=4096, rows_per_pass=4, want_logp=False):
        """One long recording [L] (float32, on the device) in BOUNDED memory: the reference CLI skips files longer than
        10 s (infer.py:201-203); the one-call path handles any length but its workspace grows with it (2.6 GiB per hour of
        audio on QuartzNet12x1).  Here the log-mel features -- whose per-feature statistics span the whole recording
        (features.py:17-30), 256 bytes per frame -- are computed in one piece, and the encoder runs over windows of
        ``chunk_frames`` OUTPUT frames, each extended by the receptive-field halo on both sides (``halo_mel_frames``;
        zero padding only where the recording really ends), ``rows_per_pass`` windows at a time as the rows of one batch.
        Frames inside a window's halo are discarded, the kept ones see exactly the inputs of the one-pass computation:
        the same log-probs and predictions BIT FOR BIT in the fp32 and 3 x bf16 arithmetics; in the default 2 x fp16 one the operand
        scale follows the row a kernel works on (a window here, the recording there), the log-probs agree within the parity
        tolerance and a frame whose two best classes are closer than that may decode differently (round-6 campaign,
        tests/devtools/fuzz_long.py: 14 444 exact cases bit-equal; 2 such frames in 7 048 f16x2 recordings).  Returns dict(ids, id_len, pred, enc_len, logp or None, workspace_bytes)."""
        from . import stages
        if wav.dim() != 1 or wav.device.type != "cuda" or wav.dtype != torch.float32:
            raise ValueError("wav must be a 1-D float32 cuda tensor")
        h = self.handle
        L = wav.shape[0]
        mel, seq = stages.melspec(h, wav[None].contiguous(), torch.tensor([L], dtype=torch.int64, device=wav.device))
        T, seq0 = mel.shape[2], int(seq[0])
        T1 = h.encoded_frames(T)
        H = self.halo_mel_frames()
        stride = 1
        for b in self._blocks:
            stride *= b["stride"]
        chunk_frames = max(int(chunk_frames), 1)
        jobs = []                                           # (mel lo, mel hi, mask length, first kept frame, kept frames)
        for o0 in range(0, T1, chunk_frames):
            o1 = min(o0 + chunk_frames, T1)
            lo = max(0, stride * o0 - H)
            lo -= lo % stride                                # windows start on the stride grid: same conv phase
            hi = min(T, stride * o1 + H)
            jobs.append((lo, hi, max(min(seq0, hi) - lo, 0), o0 - lo // stride, o1 - o0))
        c_out = self._blocks[-1]["filters"]
        V1 = len(self.labels) + 1
        logp = torch.empty((1, T1, V1), dtype=torch.float32, device=wav.device)
        ws_bytes, done = 0, 0
        for g in range(0, len(jobs), rows_per_pass):
            grp = jobs[g : g + rows_per_pass]
            W = max(hi - lo for lo, hi, *_ in grp)
            x = torch.zeros((len(grp), mel.shape[1], W), dtype=torch.float32, device=wav.device)
            for r, (lo, hi, *_rest) in enumerate(grp):
                x[r, :, : hi - lo] = mel[0, :, lo:hi]
            lens = torch.tensor([j[2] for j in grp], dtype=torch.int64, device=wav.device)
            enc, _ = stages.encoder(h, x, lens, c_out)
            ws_bytes = max(ws_bytes, h.workspace_bytes(len(grp), mel_frames=W))
            for r, (_lo, _hi, _ml, skip, keep) in enumerate(grp):
                piece = enc[r : r + 1, :, skip : skip + keep].contiguous()
                logp[:, done : done + keep] = stages.decoder(h, piece)
                done += keep
        pred = stages.greedy_argmax(logp)
        ids, id_len = stages.ctc_collapse(pred, V1 - 1)
        enc_len = torch.tensor([float(self._encoded_length(seq0))], dtype=torch.float32, device=wav.device)
        return dict(ids=ids, id_len=id_len, pred=pred, enc_len=enc_len, logp=logp if want_logp else None,
                    workspace_bytes=ws_bytes, one_pass_workspace_bytes=h.workspace_bytes(1, samples=L))

    def _encoded_length(self, seq):
        """MaskedConv1d.get_seq_len chain (jasper.py:108-111, quirk Q3) on the host: float result, truncated between convs."""
        lf = float(seq)
        first = True
        for b in self._blocks:
            k = b["kernel"] + (1 if b["kernel"] % 2 == 0 else 0)
            pad = (b["dilation"] * k) // 2 - 1 if b["dilation"] > 1 else k // 2
            for _ in range(b["repeat"]):
                convs = [(k, b["stride"], b["dilation"], pad), (1, 1, 1, 0)] if b["separable"] else [(k, b["stride"], b["dilation"], pad)]
                for kk, st, dl, pd in convs:
                    li = int(lf) if not first else int(seq)
                    first = False
                    lf = float(np.float32(np.float32(li + 2 * pd - dl * (kk - 1) - 1) / np.float32(st)) + np.float32(1.0))
        return lf

    # -- pipelined host path: pinned staging, copies on their own stream, two batches in flight
    def launch(self, signals, row_independent=False):
        """Enqueue one batch and return at once; ``.texts()`` of the returned PendingBatch waits for it.

        Two staging slots alternate: while batch k computes, batch k+1 is collated into pinned memory and its
        host->device copy runs on a separate stream (``torch.from_numpy(x).to(device)`` from pageable memory costs
        more than the whole forward pass of a 64 x 10 s batch).  int16 PCM signals cross PCIe as int16 and are scaled
        by 2^-15 on the device (segment.py:61-74) -- half the bytes, same floats.
        """
        if len(signals) == 0:
            raise ValueError("empty batch")
        if self._slots is None:
            self._slots = [_Slot(self), _Slot(self)]
            self._copy_stream = torch.cuda.Stream(self.device)
        slot = self._slots[self._launched % 2]
        self._launched += 1
        return slot.launch(signals, row_independent)


class PendingBatch:
    """Result handle of QuartzNetCTC.launch()."""

    def __init__(self, slot, batch, frames):
        self._slot, self._batch, self._frames = slot, batch, frames
        self._lock = threading.Lock()
        self._texts = None

    def done(self):
        return self._texts is not None or self._slot.ev_out.query()

    def texts(self):
        with self._lock:
            if self._texts is None:
                s, B, t1 = self._slot, self._batch, self._frames
                s.ev_out.synchronize()
                ids = s.pin_ids[: B * t1].view(B, t1).numpy()
                n = s.pin_idlen[:B].numpy()
                labels = s.eng.labels
                self._texts = ["".join(labels[c] for c in ids[b, : n[b]]) for b in range(B)]
                s.pending = None
            return self._texts


class _Slot:
    """Pinned staging + device input buffers of one in-flight batch (grown on demand, never shrunk)."""

    def __init__(self, eng):
        self.eng = eng
        self.pin = self.dev = None
        self.pin_len = self.dev_len = None
        self.pin_ids = self.pin_idlen = None
        self.out = None               # device outputs of the batch in flight (kept alive until its results are read)
        self.pending = None
        self.ev_h2d = torch.cuda.Event()
        self.ev_out = torch.cuda.Event()

    def _reserve(self, B, nbytes, t1, pcm16):
        dev = self.eng.device
        if self.pin is None or self.pin.numel() < nbytes:
            self.pin = torch.empty(nbytes, dtype=torch.uint8, pin_memory=True)
            self.dev = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        if self.pin_len is None or self.pin_len.numel() < B:
            self.pin_len = torch.empty(B, dtype=torch.int64, pin_memory=True)
            self.dev_len = torch.empty(B, dtype=torch.int64, device=dev)
            self.pin_idlen = torch.empty(B, dtype=torch.int32, pin_memory=True)
        if self.pin_ids is None or self.pin_ids.numel() < B * t1:
            self.pin_ids = torch.empty(B * t1, dtype=torch.int32, pin_memory=True)

    def launch(self, signals, row_independent=False):
        eng = self.eng
        if self.pending is not None:
            self.pending.texts()          # the slot's previous batch: fetch before its buffers are overwritten
        B = len(signals)
        lens = [len(s) for s in signals]
        L = max(lens)
        if min(lens) == 0:
            raise ValueError("empty signal in batch")
        if row_independent and min(lens) <= eng.frontend["n_fft"] // 2:
            raise ValueError(f"row-independent batching needs more than n_fft/2 = {eng.frontend['n_fft'] // 2} samples "
                             f"per signal (got {min(lens)}): an unbatched call refuses such input too")
        pcm16 = all(getattr(s, "dtype", None) == np.int16 for s in signals)
        dtype, item = (torch.int16, 2) if pcm16 else (torch.float32, 4)
        nbytes = B * L * item
        _, t1 = eng.frames(L)
        with torch.cuda.device(eng.device):
            self.ev_out.synchronize()     # batch k-2 has left the device buffers
            self._reserve(B, nbytes, t1, pcm16)
            host = self.pin[:nbytes].view(dtype).view(B, L).numpy()
            for i, s in enumerate(signals):
                if not pcm16:
                    # a MIXED batch (the serving queue merges whatever arrives): integer PCM rows are scaled here as
                    # AudioSegment does -- assigned raw they would sit 2^15 too loud next to the float rows (found by
                    # tests/devtools/stress_serving.py, round 6)
                    s = _pcm_to_float(s)
                host[i, : lens[i]] = s
                host[i, lens[i]:] = 0
            self.pin_len[:B] = torch.tensor(lens, dtype=torch.int64)
            comp = torch.cuda.current_stream(eng.device)
            with torch.cuda.stream(eng._copy_stream):
                self.dev[:nbytes].copy_(self.pin[:nbytes], non_blocking=True)
                self.dev_len[:B].copy_(self.pin_len[:B], non_blocking=True)
                self.ev_h2d.record(eng._copy_stream)
            comp.wait_event(self.ev_h2d)
            wav = self.dev[:nbytes].view(dtype).view(B, L)
            # (int16 PCM goes into the front end as it is: scaled by 2^-15 in the STFT kernel's staging load)
            self.out = eng.forward(wav, self.dev_len[:B], want_pred=False, row_independent=row_independent)
            self.pin_ids[: B * t1].copy_(self.out["ids"].view(-1), non_blocking=True)
            self.pin_idlen[:B].copy_(self.out["id_len"], non_blocking=True)
            self.ev_out.record(comp)
        self.pending = PendingBatch(self, B, t1)
        return self.pending
