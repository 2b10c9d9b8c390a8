"""HIP-backed counterparts of the reference's ASR NeuralModules (``nemo.collections.asr``).

Same class names, constructor signatures, port names / neural types and checkpoint
(``state_dict``) layout as the reference, so that infer.py-style code only changes its import:

    import viet_asr_amd.asr as nemo_asr

Every ``forward`` runs hand-written gfx950 kernels through libvasr_hip.so (stages.py); there is no
torch-eager or CPU fallback -- constructing modules and wiring the DAG works anywhere, running a
forward without a HIP device raises.
"""
import torch
import torch.nn as nn

from . import _lib, stages
from .core import (AcousticEncodedRepresentation, AudioSignal, DataLayerNM, DeviceType, LengthsType,
                   LogprobsType, MelSpectrogramType, NeuralType, NonTrainableNM, PredictionsType,
                   SpectrogramType, TrainableNM)
from .engine import blocks_from_config
from .frontend_tables import frontend_description

__all__ = ["AudioToMelSpectrogramPreprocessor", "JasperEncoder", "JasperDecoderForCTC", "GreedyCTCDecoder",
           "BeamSearchDecoderWithLM", "AudioDataLayer"]


def _no_gpu():
    return _lib.VasrError("viet-asr_amd needs a HIP device (torch.cuda.is_available() is False); "
                          "there is no CPU fallback for this path")


class AudioToMelSpectrogramPreprocessor(NonTrainableNM):
    """nemo/collections/asr/audio_preprocessing.py:212-383 (forward :78-87 -> parts/features.py:245-301)."""

    @property
    def input_ports(self):
        return {"input_signal": NeuralType(("B", "T"), AudioSignal(freq=self._sample_rate)),
                "length": NeuralType(tuple("B"), LengthsType())}

    @property
    def output_ports(self):
        return {"processed_signal": NeuralType(("B", "D", "T"), MelSpectrogramType()),
                "processed_length": NeuralType(tuple("B"), LengthsType())}

    def __init__(self, sample_rate=16000, window_size=0.02, window_stride=0.01, n_window_size=None,
                 n_window_stride=None, window="hann", normalize="per_feature", n_fft=None, preemph=0.97,
                 features=64, lowfreq=0, highfreq=None, log=True, log_zero_guard_type="add",
                 log_zero_guard_value=2 ** -24, dither=1e-5, pad_to=16, frame_splicing=1, stft_conv=False,
                 pad_value=0, mag_power=2.0):
        self._sample_rate = sample_rate
        super().__init__()
        if log_zero_guard_type not in ("add", "clamp"):
            raise ValueError(f"{self} received {log_zero_guard_type} for the log_zero_guard_type parameter. "
                             "It must be either 'add' or 'clamp'.")
        self._desc = frontend_description(dict(
            sample_rate=sample_rate, window_size=window_size, window_stride=window_stride,
            n_window_size=n_window_size, n_window_stride=n_window_stride, window=window, normalize=normalize,
            n_fft=n_fft, preemph=preemph, features=features, lowfreq=lowfreq, highfreq=highfreq, log=log,
            log_zero_guard_type=log_zero_guard_type, log_zero_guard_value=log_zero_guard_value,
            frame_splicing=frame_splicing, stft_conv=stft_conv, pad_value=pad_value, mag_power=mag_power))
        # dither (features.py:250-251): `x += dither * torch.randn_like(x)` -- the SAME torch call on the same device tensor, in
        # place like the reference (SURVEY section 8b "ownership"), so that under one torch.manual_seed this module and the
        # reference's draw the same noise on the same device; infer.py:89 forces 0.  pad_to: FilterbankFeatures zero-pads T up
        # to a multiple of self.pad_to in training mode -- the mode the executor leaves this NonTrainableNM in, quirk Q1 -- and
        # of 16 in eval mode (features.py:292-300); infer.py:90 sets 0 = no padding.  A non-zero pad_to is applied here the way
        # the reference's (training-mode) branch does it: extra all-zero frames appended on the device.
        if dither is not None and dither < 0:
            raise ValueError(f"dither must be >= 0, got {dither!r}")
        if pad_to == "max":
            # features.py:209 evaluates `pad_to > 0` in the constructor: with the string the docstring advertises
            # (audio_preprocessing.py:261-262, features.py:295-296) the reference itself raises this TypeError
            raise TypeError("'>' not supported between instances of 'str' and 'int' (pad_to='max': the reference's own "
                            "constructor fails at parts/features.py:209)")
        if pad_to is not None and (int(pad_to) != pad_to or pad_to < 0):
            raise ValueError(f"pad_to must be a non-negative integer, got {pad_to!r}")
        self.dither, self.pad_to = float(dither or 0.0), int(pad_to or 0)
        self.win_length, self.hop_length = self._desc["win_length"], self._desc["hop_length"]
        self._handle = None

    def _get_handle(self):
        if self._handle is None:
            if not torch.cuda.is_available():
                raise _no_gpu()
            self._handle = _lib.Handle(frontend=self._desc)
            self._handle.finalize()
        return self._handle

    @property
    def filter_banks(self):
        return torch.from_numpy(self._desc["filterbank"]).unsqueeze(0)

    def get_seq_len(self, seq_len):
        return torch.ceil(seq_len.float() / self.hop_length).to(dtype=torch.long)

    def forward(self, input_signal, length):
        if self.dither > 0:
            input_signal += self.dither * torch.randn_like(input_signal)      # features.py:250-251
        mel, seq = stages.melspec(self._get_handle(), input_signal, length)
        if self.pad_to > 0 and mel.shape[-1] % self.pad_to:
            mel = torch.nn.functional.pad(mel, (0, self.pad_to - mel.shape[-1] % self.pad_to), value=self._desc.get("pad_value", 0.0))
        return mel, seq


class _MaskedConvParams(nn.Module):
    """Parameter container with the reference's MaskedConv1d key layout (``.conv.weight``)."""

    def __init__(self, cin, cout, k, groups=1):
        super().__init__()
        self.conv = nn.Conv1d(cin, cout, k, groups=groups, bias=False)


class _JasperBlockParams(nn.Module):
    """ModuleList skeleton of one JasperBlock (parts/jasper.py:214-288): same indices, parameters only."""

    def __init__(self, inplanes, planes, repeat, kernel, separable, residual):
        super().__init__()
        layers, c = [], inplanes
        for r in range(repeat):
            if separable:
                layers += [_MaskedConvParams(c, c, kernel, groups=c), _MaskedConvParams(c, planes, 1)]
            else:
                layers += [_MaskedConvParams(c, planes, kernel)]
            layers.append(nn.BatchNorm1d(planes, eps=1e-3, momentum=0.1))
            if r != repeat - 1:
                layers += [nn.Identity(), nn.Identity()]      # activation, dropout slots
            c = planes
        self.mconv = nn.ModuleList(layers)
        self.res = None
        if residual:
            self.res = nn.ModuleList([nn.ModuleList([_MaskedConvParams(inplanes, planes, 1),
                                                     nn.BatchNorm1d(planes, eps=1e-3, momentum=0.1)])])


def _init_weights(m, mode="xavier_uniform"):
    """parts/jasper.py:28-49."""
    if isinstance(m, (nn.Conv1d, nn.Linear)):
        init = {"xavier_uniform": lambda w: nn.init.xavier_uniform_(w, gain=1.0),
                "xavier_normal": lambda w: nn.init.xavier_normal_(w, gain=1.0),
                "kaiming_uniform": lambda w: nn.init.kaiming_uniform_(w, nonlinearity="relu"),
                "kaiming_normal": lambda w: nn.init.kaiming_normal_(w, nonlinearity="relu")}
        if mode not in init:
            raise ValueError("Unknown Initialization mode: {0}".format(mode))
        init[mode](m.weight)
    elif isinstance(m, nn.BatchNorm1d):
        m.running_mean.zero_()
        m.running_var.fill_(1)
        m.num_batches_tracked.zero_()
        nn.init.ones_(m.weight)
        nn.init.zeros_(m.bias)


class _HipWeights:
    """Mixin: keeps a libvasr handle in sync with the module's state_dict (re-packed after any load)."""

    def _invalidate(self):
        self._handle = None

    def load_state_dict(self, state_dict, strict=True):
        r = nn.Module.load_state_dict(self, state_dict, strict=strict)
        self._invalidate()
        return r


class JasperEncoder(_HipWeights, TrainableNM):
    """nemo/collections/asr/jasper.py:17-204."""

    @property
    def input_ports(self):
        return {"audio_signal": NeuralType(("B", "D", "T"), SpectrogramType()),
                "length": NeuralType(tuple("B"), LengthsType())}

    @property
    def output_ports(self):
        return {"outputs": NeuralType(("B", "D", "T"), AcousticEncodedRepresentation()),
                "encoded_lengths": NeuralType(tuple("B"), LengthsType())}

    def __init__(self, jasper, activation, feat_in, normalization_mode="batch", residual_mode="add", norm_groups=-1,
                 conv_mask=True, frame_splicing=1, init_mode="xavier_uniform"):
        super().__init__()
        if activation not in ("hardtanh", "relu", "selu"):
            raise KeyError(activation)
        if activation != "relu" or normalization_mode != "batch" or residual_mode != "add" or not conv_mask \
                or frame_splicing != 1:
            raise NotImplementedError("implemented: activation='relu', normalization_mode='batch', "
                                      "residual_mode='add', conv_mask=True, frame_splicing=1")
        self._blocks = blocks_from_config(jasper)
        self._feat_in = feat_in * frame_splicing
        for b in self._blocks:
            if b["stride"] > 1 and b["dilation"] > 1:
                raise ValueError("Only stride OR dilation may be greater than 1")   # parts/jasper.py:61-62
        layers, c = [], self._feat_in
        for b in self._blocks:
            k = b["kernel"] + (1 if b["kernel"] % 2 == 0 else 0)
            layers.append(_JasperBlockParams(c, b["filters"], b["repeat"], k, bool(b["separable"]), bool(b["residual"])))
            c = b["filters"]
        self.encoder = nn.Sequential(*layers)
        self._c_out = c
        self.apply(lambda m: _init_weights(m, mode=init_mode))
        self._handle = None

    def _get_handle(self):
        if self._handle is None:
            if not torch.cuda.is_available():
                raise _no_gpu()
            h = _lib.Handle(feat_in=self._feat_in, blocks=self._blocks)
            h.load_state_dict(self.state_dict())
            h.finalize()
            self._handle = h
        return self._handle

    def forward(self, audio_signal, length=None):
        if length is None:
            length = torch.full((audio_signal.shape[0],), audio_signal.shape[2], dtype=torch.int64,
                                device=audio_signal.device)
            return stages.encoder(self._get_handle(), audio_signal, length, self._c_out)[0]
        return stages.encoder(self._get_handle(), audio_signal, length, self._c_out)


class JasperDecoderForCTC(_HipWeights, TrainableNM):
    """nemo/collections/asr/jasper.py:207-254."""

    @property
    def input_ports(self):
        return {"encoder_output": NeuralType(("B", "D", "T"), AcousticEncodedRepresentation())}

    @property
    def output_ports(self):
        return {"output": NeuralType(("B", "T", "D"), LogprobsType())}

    def __init__(self, feat_in, num_classes, init_mode="xavier_uniform"):
        super().__init__()
        self._feat_in = feat_in
        self._num_classes = num_classes + 1          # + blank
        self.decoder_layers = nn.Sequential(nn.Conv1d(self._feat_in, self._num_classes, kernel_size=1, bias=True))
        self.apply(lambda m: _init_weights(m, mode=init_mode))
        self._handle = None

    def _get_handle(self):
        if self._handle is None:
            if not torch.cuda.is_available():
                raise _no_gpu()
            h = _lib.Handle(dec_feat_in=self._feat_in, num_classes=self._num_classes)
            h.load_state_dict(self.state_dict())
            h.finalize()
            self._handle = h
        return self._handle

    def forward(self, encoder_output):
        return stages.decoder(self._get_handle(), encoder_output)


class GreedyCTCDecoder(TrainableNM):
    """nemo/collections/asr/greedy_ctc_decoder.py:9-36."""

    @property
    def input_ports(self):
        return {"log_probs": NeuralType(("B", "T", "D"), LogprobsType())}

    @property
    def output_ports(self):
        return {"predictions": NeuralType(("B", "T"), PredictionsType())}

    def __init__(self):
        super().__init__()

    def forward(self, log_probs):
        return stages.greedy_argmax(log_probs)


class BeamSearchDecoderWithLM(NonTrainableNM):
    """nemo/collections/asr/beam_search_decoder.py:14-102 (pyctcdecode-backed in the reference).

    The prefix beam search (with optional n-gram LM) lives in beam.py; unlike the reference it
    accepts any batch size and stays on the device.  ``forward`` returns the best string for B == 1
    (what infer.py consumes: evaluated_tensors[0][0]) and a list of strings otherwise.
    """

    @property
    def input_ports(self):
        return {"log_probs": NeuralType(("B", "T", "D"), LogprobsType()),
                "log_probs_length": NeuralType(tuple("B"), LengthsType())}

    @property
    def output_ports(self):
        return {"predictions": NeuralType(("B", "T"), PredictionsType())}

    def __init__(self, lm_path, vocab, beam_width, alpha, beta, num_cpus=1, cutoff_prob=1.0, cutoff_top_n=40,
                 input_tensor=True, allow_missing_lm=False, unigrams="auto"):
        """unigrams (net-new): "auto" = what build_ctcdecoder does for lm_path's suffix -- the ARPA's own unigram list + character
        trie for "*.arpa", none otherwise --, None = no unigram list on any file (the behaviour the reference got from its
        `.binary`), or a list of words (viet_asr_amd/beam.py)."""
        super().__init__()
        if self._factory is not None and self._factory.world_size > 1:
            raise ValueError("BeamSearchDecoderWithLM does not run in distributed mode")   # :79-80
        from .beam import BeamSearchDecoder
        self.vocab, self.beam_width = list(vocab), beam_width
        # an lm_path that cannot be read (KenLM binaries, a missing file) raises unless allow_missing_lm (beam.LM_HELP)
        self.decoder = BeamSearchDecoder(self.vocab, lm_path=lm_path, alpha=alpha, beta=beta, allow_missing_lm=allow_missing_lm,
                                         unigrams=unigrams)
        self.num_cpus, self.cutoff_prob, self.cutoff_top_n, self.input_tensor = num_cpus, cutoff_prob, cutoff_top_n, \
            input_tensor

    def forward(self, log_probs, log_probs_length=None):
        # Batch 1 (all the reference accepts, :96) searches every frame, like pyctcdecode on log_probs[0].  In a padded
        # batch the shorter rows' trailing frames hold the head's output on masked input, not blanks: each row is
        # searched over its own log_probs_length[b] frames (float lengths from the encoder, quirk Q3, truncated).
        frames = None
        if log_probs.shape[0] > 1 and log_probs_length is not None:
            frames = torch.as_tensor(log_probs_length).to(torch.float64).floor().clamp(0, log_probs.shape[1]).to(torch.int32)
        texts = self.decoder.decode_batch(log_probs, beam_width=self.beam_width, frames=frames)
        return texts[0] if len(texts) == 1 else texts


class AudioDataLayer(DataLayerNM):
    """Batched counterpart of infer.py:16-54: a one-shot iterator over (audio_signal [B,L], a_sig_length [B]).

    ``set_signal(x)`` keeps the reference's single-utterance call; ``set_batch(list)`` zero-pads to the
    longest utterance like ``seq_collate_fn`` (parts/dataset.py:14-53).
    """

    @property
    def output_ports(self):
        return {"audio_signal": NeuralType(("B", "T"), AudioSignal(freq=self._sample_rate)),
                "a_sig_length": NeuralType(tuple("B"), LengthsType())}

    def __init__(self, sample_rate):
        super().__init__()
        self._sample_rate = sample_rate
        self.output = False
        self.signal = self.signal_shape = None

    def set_signal(self, signal):
        """The reference's single-utterance call (infer.py:29-33): a batch of one."""
        import numpy as np
        self.set_batch([np.asarray(signal, dtype=np.float32).reshape(-1)])

    def set_batch(self, signals):
        import numpy as np
        lens = np.array([len(s) for s in signals], dtype=np.int64)
        batch = np.zeros((len(signals), int(lens.max())), dtype=np.float32)
        for i, s in enumerate(signals):
            batch[i, : len(s)] = np.asarray(s, dtype=np.float32)
        self.signal, self.signal_shape, self.output = batch, lens, True

    def __iter__(self):
        return self

    def __next__(self):
        if not self.output:
            raise StopIteration
        self.output = False
        return torch.as_tensor(self.signal, dtype=torch.float32), torch.as_tensor(self.signal_shape, dtype=torch.int64)

    def __len__(self):
        return 1

    @property
    def dataset(self):
        return None

    @property
    def data_iterator(self):
        return self
