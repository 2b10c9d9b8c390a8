"""VietASR: counterpart of the reference's infer.py:57-171 on the HIP modules.

Same constructor and ``transcribe(audio_signal) -> str``.  Differences, all additive:
  * ``decoder="greedy"`` (the variant infer.py:113 has commented out) next to the reference's
    ``"beam"`` wiring (infer.py:132-139, 159-160);
  * ``transcribe_batch(list_of_signals)`` runs the fused one-call path (engine.QuartzNetCTC);
  * config files in either spelling (``AudioToMelSpectrogramPreprocessor`` / ``AudioPreprocessing``)
    or a builtin model name are accepted.
Audio decoding / resampling (librosa.load in the reference CLI, infer.py:200) stays with the caller.
"""
import os

import numpy as np
import torch

from . import asr as nemo_asr
from . import configs
from .core import DeviceType, NeuralModuleFactory
from .engine import QuartzNetCTC
from .helpers import post_process_predictions


class VietASR:
    def __init__(self, config_file, encoder_checkpoint, decoder_checkpoint, device="gpu", lm_path=None,
                 beam_width=20, lm_alpha=0.5, lm_beta=1.5, decoder="beam", allow_missing_lm=False, lm_unigrams="auto"):
        if os.path.exists(str(config_file)):
            model_definition = configs.load_model_definition(config_file)
        else:
            model_definition = configs.builtin(config_file)        # raises ValueError for unknown names
        assert os.path.exists(encoder_checkpoint), f"encoder checkpoint not found: {encoder_checkpoint}"
        assert os.path.exists(decoder_checkpoint), f"decoder checkpoint not found: {decoder_checkpoint}"
        pre = model_definition["AudioToMelSpectrogramPreprocessor"]
        pre["dither"] = 0          # infer.py:89
        pre["pad_to"] = 0          # infer.py:90
        if device != "gpu" or not torch.cuda.is_available():
            raise RuntimeError("viet-asr_amd runs on a HIP device only (device='gpu'); there is no CPU path")
        self.model_definition = model_definition
        self.labels = labels = model_definition["labels"]
        self.neural_factory = NeuralModuleFactory(placement=DeviceType.GPU)
        self.data_layer = nemo_asr.AudioDataLayer(sample_rate=pre["sample_rate"])
        self.preprocessor = nemo_asr.AudioToMelSpectrogramPreprocessor(**pre)
        self.encoder = nemo_asr.JasperEncoder(feat_in=pre["features"], **model_definition["JasperEncoder"])
        self.decoder = nemo_asr.JasperDecoderForCTC(
            feat_in=model_definition["JasperEncoder"]["jasper"][-1]["filters"], num_classes=len(labels))
        self.encoder.restore_from(encoder_checkpoint)
        self.decoder.restore_from(decoder_checkpoint)

        audio_signal, audio_signal_len = self.data_layer()
        processed_signal, processed_signal_len = self.preprocessor(input_signal=audio_signal, length=audio_signal_len)
        encoded, encoded_len = self.encoder(audio_signal=processed_signal, length=processed_signal_len)
        log_probs = self.decoder(encoder_output=encoded)
        self.mode = decoder
        if decoder == "greedy":
            self.greedy = nemo_asr.GreedyCTCDecoder()
            self.infer_tensors = [self.greedy(log_probs=log_probs)]
        elif decoder == "beam":
            # a given lm_path that is missing or not ARPA text is an error unless allow_missing_lm (round 5; beam.LM_HELP)
            self.beam = nemo_asr.BeamSearchDecoderWithLM(vocab=labels, beam_width=beam_width, alpha=lm_alpha,
                                                         beta=lm_beta, lm_path=lm_path, allow_missing_lm=allow_missing_lm,
                                                         unigrams=lm_unigrams,   # "auto": pyctcdecode's rule for the path's suffix
                                                         num_cpus=max(1, os.cpu_count()))
            self.infer_tensors = [self.beam(log_probs=log_probs, log_probs_length=encoded_len)]
        else:
            raise ValueError(f"decoder must be 'greedy' or 'beam', got {decoder!r}")
        self._fused = None

    def _to_model_rate(self, audio_signal, sample_rate):
        """The reference CLI resamples with librosa.load(sr=16000) (infer.py:200); here on the device."""
        rate = self.model_definition["AudioToMelSpectrogramPreprocessor"]["sample_rate"]
        x = np.asarray(audio_signal)
        if x.dtype.kind == "i":                     # integer PCM: AudioSegment scaling (segment.py:61-74)
            x = x.astype(np.float32) * (1.0 / 2 ** (8 * x.dtype.itemsize - 1))
        x = x.astype(np.float32)
        if sample_rate is None or int(sample_rate) == int(rate):
            return x
        from . import audio
        y, n = audio.resample(torch.from_numpy(x)[None].cuda(), torch.tensor([len(x)], device="cuda"), sample_rate, rate)
        return y[0, : int(n[0])].cpu().numpy()

    def transcribe(self, audio_signal, sample_rate=None):
        audio_signal = self._to_model_rate(audio_signal, sample_rate)
        self.data_layer.set_signal(audio_signal)
        evaluated = self.neural_factory.infer(tensors=self.infer_tensors, verbose=False)
        if self.mode == "greedy":
            return post_process_predictions(evaluated[0], self.labels)[0]
        return evaluated[0][0]

    def _fused_engine(self):
        if self._fused is None:
            self._fused = QuartzNetCTC(self.model_definition, self.encoder.state_dict(), self.decoder.state_dict())
        return self._fused

    def _batch_signals(self, signals, sample_rate):
        rate = self.model_definition["AudioToMelSpectrogramPreprocessor"]["sample_rate"]
        if sample_rate is None or int(sample_rate) == int(rate):
            if all(getattr(s, "dtype", None) == np.int16 for s in signals):
                return signals                      # int16 PCM goes to the device as it is (scaled there)
        return [self._to_model_rate(s, sample_rate) for s in signals]

    def transcribe_batch(self, signals, sample_rate=None, row_independent=False, decoder="greedy"):
        """Transcripts of a list of 1-D signals through the fused one-call path.  row_independent=True: every
        transcript is what the signal alone would give (engine.QuartzNetCTC.forward); default: the reference's
        padded-batch semantics.  decoder="beam" (instances built with decoder="beam" only) runs this instance's beam
        search + LM over the batch instead of the greedy collapse."""
        if decoder == "greedy":
            return self._fused_engine().transcribe(self._batch_signals(signals, sample_rate), row_independent)
        if decoder != "beam" or self.mode != "beam":
            raise ValueError("decoder must be 'greedy', or 'beam' on an instance constructed with decoder='beam'")
        sigs = [self._to_model_rate(s, sample_rate) for s in signals]
        return self._fused_engine().transcribe_beam(sigs, self.beam.decoder, self.beam.beam_width, row_independent)

    def transcribe_manifest(self, manifest_filepath, batch_size=64, row_independent=True):
        """Greedy transcripts of every entry of a NeMo JSON-lines manifest (parts/manifest.py:21-94), in manifest order,
        plus the word error rate against the entries' ``text`` (None when no entry has one).

        The reference CLI walks a directory one file at a time (infer.py:194-206).  Here the entries are sorted by
        duration, cut into batches of ``batch_size`` and pushed through the pipelined engine two batches at a time
        (engine.QuartzNetCTC.launch); with row_independent=True (default) every transcript is what ``transcribe`` returns
        for that file alone.  PCM WAV only; files at another rate are resampled on the device."""
        import json
        from . import audio
        from .data_layer import word_error_rate
        entries = []
        for path in str(manifest_filepath).split(","):
            with open(path, encoding="utf-8") as f:
                entries += [json.loads(line) for line in f if line.strip()]
        order = sorted(range(len(entries)), key=lambda i: float(entries[i].get("duration", 0.0)))
        hyps, pending = [None] * len(entries), []

        def collect(job):
            idx, handle = job
            for i, t in zip(idx, handle.texts()):
                hyps[i] = t

        for lo in range(0, len(order), batch_size):
            idx = order[lo : lo + batch_size]
            sigs = []
            for i in idx:
                x, sr = audio.read_wav(entries[i]["audio_filepath"])
                sigs.append(self._to_model_rate(x, sr))
            pending.append((idx, self._fused_engine().launch(sigs, row_independent)))
            if len(pending) == 2:
                collect(pending.pop(0))
        for job in pending:
            collect(job)
        refs = [e.get("text") for e in entries]
        scored = [i for i, r in enumerate(refs) if r]
        wer = word_error_rate([hyps[i] for i in scored], [refs[i] for i in scored]) if scored else None
        return hyps, wer

    def launch_batch(self, signals, sample_rate=None, row_independent=False):
        """Asynchronous ``transcribe_batch``: returns a handle at once, ``.texts()`` waits (engine.QuartzNetCTC.launch)."""
        return self._fused_engine().launch(self._batch_signals(signals, sample_rate), row_independent)
