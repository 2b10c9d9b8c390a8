"""Per-stage wrappers over the C ABI: torch tensors in, torch tensors out, on the current stream.

Each function is the forward of one reference NeuralModule (ports and dtypes as in
SURVEY.md §8b); asr.py's module classes call these.  Result tensors are allocated here with
torch (caching allocator); the library only fills them.
"""
import torch

from . import _lib


def _st():
    return torch.cuda.current_stream().cuda_stream


def _need_cuda(*ts):
    for t in ts:
        if t.device.type != "cuda":
            raise _lib.VasrError("viet-asr_amd kernels need HIP-resident tensors (got a CPU tensor); "
                                 "there is no CPU fallback for this path")


# One workspace per (device, STREAM): the library is re-entrant per handle + stream (include/vasr.h), so two host threads
# driving per-module calls on two streams must not be handed the same scratch memory (rounds 1-5 kept one per device).
# Allocated while its stream is current, so the caching allocator's stream-ordered reuse covers a replaced (grown) buffer.
_ws_cache = {}
_ws_lock = __import__("threading").Lock()


def _workspace(device, nbytes):
    key = (device, _st())
    with _ws_lock:
        ws = _ws_cache.get(key)
        if ws is None or ws.numel() < nbytes:
            _ws_cache[key] = None
            ws = torch.empty(max(nbytes, 256), dtype=torch.uint8, device=device)
            _ws_cache[key] = ws
        return ws


def melspec(handle, input_signal, length):
    """AudioToMelSpectrogramPreprocessor.forward -> (processed_signal [B,64,T] f32, processed_length [B] i64)."""
    _need_cuda(input_signal, length)
    x = input_signal.to(torch.float32).contiguous()
    ln = length.to(torch.int64).contiguous()
    B, L = x.shape
    T = handle.mel_frames(L)
    mel = torch.empty((B, 64, T), dtype=torch.float32, device=x.device)
    seq = torch.empty((B,), dtype=torch.int64, device=x.device)
    _lib.check(_lib.lib().vasr_melspec_f32(handle.h, x.data_ptr(), ln.data_ptr(), B, L, mel.data_ptr(),
                                           seq.data_ptr(), _st()))
    return mel, seq


def encoder(handle, audio_signal, length, c_out):
    """JasperEncoder.forward -> (outputs [B,C,T'] f32, encoded_lengths [B] f32)."""
    _need_cuda(audio_signal, length)
    x = audio_signal.to(torch.float32).contiguous()
    ln = length.to(torch.int64).contiguous()
    B, _, T = x.shape
    T1 = handle.encoded_frames(T)
    out = torch.empty((B, c_out, T1), dtype=torch.float32, device=x.device)
    enc_len = torch.empty((B,), dtype=torch.float32, device=x.device)
    ws = _workspace(x.device, handle.workspace_bytes(B, mel_frames=T))
    _lib.check(_lib.lib().vasr_encoder_f32(handle.h, x.data_ptr(), ln.data_ptr(), B, T, out.data_ptr(),
                                           enc_len.data_ptr(), ws.data_ptr(), ws.numel(), _st()))
    return out, enc_len


def decoder(handle, encoder_output):
    """JasperDecoderForCTC.forward -> log_probs [B,T',V+1] f32."""
    _need_cuda(encoder_output)
    x = encoder_output.to(torch.float32).contiguous()
    B, C, T1 = x.shape
    V = handle.num_classes
    ld = int(_lib.lib().vasr_padded_frames(T1))
    need = ((B * C * ld * 4 + 255) // 256) * 256 + B * V * ld * 4
    need = ((need + 255) // 256) * 256 + B * 1024       # + the maxima table of the port tensor (fp16-split head, vasr.h)
    ws = _workspace(x.device, need)
    logp = torch.empty((B, T1, V), dtype=torch.float32, device=x.device)
    _lib.check(_lib.lib().vasr_decoder_logsoftmax_f32(handle.h, x.data_ptr(), B, T1, logp.data_ptr(),
                                                      ws.data_ptr(), ws.numel(), _st()))
    return logp


def greedy_argmax(log_probs):
    """GreedyCTCDecoder.forward -> predictions [B,T'] i64."""
    _need_cuda(log_probs)
    x = log_probs.to(torch.float32).contiguous()
    B, T1, V = x.shape
    pred = torch.empty((B, T1), dtype=torch.int64, device=x.device)
    _lib.check(_lib.lib().vasr_greedy_argmax(x.data_ptr(), B, T1, V, pred.data_ptr(), _st()))
    return pred


def ctc_collapse(predictions, blank_id):
    """Device side of __ctc_decoder_predictions_tensor -> (ids [B,T'] i32, id_len [B] i32)."""
    _need_cuda(predictions)
    p = predictions.to(torch.int64).contiguous()
    B, T1 = p.shape
    ids = torch.empty((B, T1), dtype=torch.int32, device=p.device)
    n = torch.empty((B,), dtype=torch.int32, device=p.device)
    _lib.check(_lib.lib().vasr_ctc_collapse(p.data_ptr(), B, T1, int(blank_id), ids.data_ptr(), n.data_ptr(), _st()))
    return ids, n
