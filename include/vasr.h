/*
 * vasr.h -- C ABI of libvasr_hip.so: the MI355X (gfx950) implementation of the
 * viet-asr infer.py hot path.
 *
 * Nothing like this exists in the reference (it is Python on ATen ops only); each entry
 * point below names the reference interface whose arithmetic it replaces, paths under
 * /root/reference.  The boundary carries plain pointers, sizes and a hipStream_t: no torch
 * types.  The library never allocates or frees caller-visible result buffers; weights are
 * copied into a library-owned handle at vasr_load_weight()/vasr_finalize().
 *
 * All pointers named d_* are DEVICE pointers (HBM); h_* are HOST pointers.
 * All functions return 0 on success or a negative vasr_status; vasr_last_error() gives a
 * thread-local human readable message for the last failure.
 * Re-entrancy: calls on distinct handles, or on one handle with distinct workspaces and
 * streams, may run concurrently; there is no global mutable state besides the error string.
 * Round 6 made that true ON THE DEVICE as well: next to 16-bit matrix kernels of another stream -- a second handle of this
 * library, or anybody else's GEMM -- packed-FP32 and FP64 vector arithmetic of a small kernel returned wrong values on MI355X
 * (profiles/r06_concurrency.txt); the front end no longer contains the first and runs the second alone on its compute unit.
 * tests/devtools/stress_attack.py and stress_threads.py are the harnesses: every entry point next to torch's fp16 bmm, and
 * N host threads on N streams, each result against the idle-device one bit for bit.
 */
#ifndef VASR_H_
#define VASR_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* The library is built with -fvisibility=hidden: the functions marked VASR_API are its ONLY dynamic symbols. */
#if defined(__GNUC__) || defined(__clang__)
#define VASR_API __attribute__((visibility("default")))
#else
#define VASR_API
#endif

/* Binary interface number of this header; vasr_abi_version() returns the one the library was built from.  Bumped whenever
 * a signature or struct layout changes (4: vasr_profile_end reports five kernel classes with flops / bytes per class --
 * a caller built against the four-class form would be written past its arrays; 5: vasr_lm_create takes 16-byte table
 * entries with power-of-two capacities; 6: vasr_lm_create takes the character trie of pyctcdecode's unigram set, the
 * vocabulary entries carry a set-membership flag, vasr_resample_f32 emits ceil(len * ratio) samples,
 * vasr_frontend_desc ends in log_guard_clamp and knows normalize = 2; 7: + vasr_transcribe_greedy_pcm16). */
#define VASR_ABI_VERSION 7

typedef struct vasr_handle vasr_handle;
typedef void* vasr_stream; /* hipStream_t */

typedef enum {
  VASR_OK = 0,
  VASR_ERR_INVALID = -1,     /* bad argument (Python side raises ValueError) */
  VASR_ERR_STATE = -2,       /* call order: weight missing, not finalized ... */
  VASR_ERR_HIP = -3,         /* HIP runtime failure */
  VASR_ERR_WORKSPACE = -4,   /* workspace too small */
  VASR_ERR_UNSUPPORTED = -5  /* configuration the kernels do not cover */
} vasr_status;

/* One JasperBlock (nemo/collections/asr/parts/jasper.py:175-288), as the YAML spells it
 * (configs/quartznet12x1_vi.yaml:25-165). */
typedef struct {
  int32_t filters;
  int32_t repeat;
  int32_t kernel;
  int32_t stride;
  int32_t dilation;
  int32_t residual;  /* 0/1 */
  int32_t separable; /* 0/1 */
} vasr_block_desc;

/* Front end = FilterbankFeatures.__init__ (parts/features.py:113-236) with the knobs the
 * path uses: dither 0, pad_to 0 (infer.py:89-90), stft_conv false, mag_power 2, frame_splicing 1; log guard "add" or
 * "clamp" (:269-274), normalisation per_feature, all_features or none (:17-46). */
typedef struct {
  int32_t sample_rate;  /* 16000 */
  int32_t n_fft;        /* 512 (only 512 is implemented) */
  int32_t win_length;   /* 320 */
  int32_t hop_length;   /* 160 */
  int32_t n_mels;       /* 64 (only 64 is implemented) */
  float preemph;        /* 0.97; <0 disables */
  float log_guard;      /* 2^-24 */
  int32_t normalize;    /* 1 = per_feature (per utterance and mel bin), 2 = all_features (one mean / std per utterance over
                           every bin and frame, features.py:31-39), 0 = none */
  const float* h_window;     /* [win_length] or NULL -> symmetric hann */
  const float* h_filterbank; /* [n_mels][n_fft/2+1] row-major, REQUIRED */
  int32_t log_guard_clamp;   /* 0 = log(x + log_guard) (log_zero_guard_type "add", the shipped configs), 1 = log(max(x,
                                log_guard)) ("clamp", features.py:272-273).  (ABI 6: appended) */
} vasr_frontend_desc;

typedef struct {
  const vasr_frontend_desc* frontend; /* NULL -> handle has no front end */
  int32_t feat_in;                    /* encoder input channels (64) */
  int32_t n_blocks;                   /* 0 -> handle has no encoder */
  const vasr_block_desc* blocks;
  int32_t dec_feat_in;                /* 1024; 0 -> handle has no CTC head */
  int32_t num_classes;                /* V+1, blank = V is the last class */
} vasr_model_desc;

/* ---- life cycle ------------------------------------------------------------------- */
VASR_API int vasr_create(const vasr_model_desc* desc, vasr_handle** out);
VASR_API void vasr_destroy(vasr_handle* h);

/* Replaces TrainableNM.restore_from -> load_state_dict (nemo/backends/pytorch/nm.py:97-103):
 * feed every float tensor of the module state_dict under its reference key, e.g.
 * "encoder.3.mconv.1.conv.weight", "encoder.3.mconv.2.running_var",
 * "encoder.3.res.0.0.conv.weight", "decoder_layers.0.bias".
 * Keys ending in "num_batches_tracked" are accepted and ignored. */
VASR_API int vasr_load_weight(vasr_handle* h, const char* key, const float* h_data, const int64_t* shape, int ndim);

/* Checks that every tensor arrived, folds eval-mode BatchNorm1d(eps=1e-3)
 * (parts/jasper.py:392) into per-channel (scale, shift), packs the 1x1-conv weights
 * K-major for the MFMA kernels and uploads everything.  Needed before any compute call. */
VASR_API int vasr_finalize(vasr_handle* h);

/* ---- shapes ------------------------------------------------------------------------ */
/* T = 1 + L / hop (torch.stft center=True, parts/features.py:181-188). */
VASR_API int64_t vasr_mel_frames(const vasr_handle* h, int64_t samples);
/* T' after every strided block: floor((T + 2p - d(K-1) - 1)/s) + 1 (parts/jasper.py:108-111). */
VASR_API int64_t vasr_encoded_frames(const vasr_handle* h, int64_t mel_frames);
/* Scratch needed by vasr_encoder_f32 / vasr_decoder_* / vasr_transcribe_greedy_f32 for a
 * batch of B utterances padded to `samples` (or, if samples == 0, to mel_frames). */
VASR_API size_t vasr_workspace_bytes(const vasr_handle* h, int batch, int64_t samples, int64_t mel_frames);

/* ---- the path, stage by stage (each = one NeuralModule forward) ---------------------- */

/* AudioToMelSpectrogramPreprocessor.forward == FilterbankFeatures.forward
 * (audio_preprocessing.py:78-87, parts/features.py:245-301).
 *   d_wav [B][L] f32 (rows zero padded), d_len [B] i64
 *   -> d_mel [B][n_mels][T] f32 contiguous, d_seq [B] i64 = ceil(len/hop) */
VASR_API int vasr_melspec_f32(vasr_handle* h, const float* d_wav, const int64_t* d_len, int batch, int64_t samples,
                     float* d_mel, int64_t* d_seq, vasr_stream stream);

/* JasperEncoder.forward (jasper.py:198-204; JasperBlock.forward parts/jasper.py:408-448;
 * MaskedConv1d.forward parts/jasper.py:113-132).
 *   d_mel [B][feat_in][T] f32 contiguous, d_seq [B] i64
 *   -> d_enc [B][C_last][T'] f32 contiguous, d_enc_len [B] f32 (quirk Q3: float lengths) */
VASR_API int vasr_encoder_f32(vasr_handle* h, const float* d_mel, const int64_t* d_seq, int batch, int64_t mel_frames,
                     float* d_enc, float* d_enc_len, void* d_workspace, size_t workspace_bytes,
                     vasr_stream stream);

/* JasperDecoderForCTC.forward (jasper.py:253-254): conv1x1+bias -> transpose -> log_softmax.
 *   d_enc [B][dec_feat_in][T'] f32 contiguous -> d_logp [B][T'][V+1] f32
 * Workspace: align256(B * dec_feat_in * ld * 4) + B * (V+1) * ld * 4 bytes with ld = vasr_padded_frames(T'); with
 * align256(that) + B * 1024 bytes the maxima of the port tensor are taken first and the head GEMM runs in the handle's GEMM mode
 * like the fused path's (the same bits as vasr_transcribe_greedy_f32 on the same encoder output); with less, mode 3 falls back
 * to the 3 x bf16 form (same tolerance). */
VASR_API int vasr_decoder_logsoftmax_f32(vasr_handle* h, const float* d_enc, int batch, int64_t enc_frames,
                                float* d_logp, void* d_workspace, size_t workspace_bytes, vasr_stream stream);

/* GreedyCTCDecoder.forward (greedy_ctc_decoder.py:33-36): argmax(-1), first max wins.
 *   d_logp [B][T'][V+1] f32 -> d_pred [B][T'] i64 */
VASR_API int vasr_greedy_argmax(const float* d_logp, int batch, int64_t frames, int num_classes, int64_t* d_pred,
                       vasr_stream stream);

/* __ctc_decoder_predictions_tensor inner loop (helpers.py:20-31): drop repeats and blanks over
 * ALL frames (quirk Q4).  d_pred [B][T'] i64 -> d_ids [B][T'] i32 (compacted), d_id_len [B] i32 */
VASR_API int vasr_ctc_collapse(const int64_t* d_pred, int batch, int64_t frames, int blank_id, int32_t* d_ids,
                      int32_t* d_id_len, vasr_stream stream);

/* ---- the whole path in one call (the fast path bench.py times) ------------------------ */
/* wav -> mel -> encoder -> CTC head -> log-softmax/argmax -> collapse, all intermediates in
 * the workspace (padded time stride, no port tensors materialised).
 *   d_pred   [B][T'] i64  (may be NULL)          d_ids [B][T'] i32, d_id_len [B] i32
 *   d_logp   [B][T'][V+1] f32 (may be NULL: greedy only needs the argmax)
 *   d_enc_len [B] f32 (may be NULL) */
VASR_API int vasr_transcribe_greedy_f32(vasr_handle* h, const float* d_wav, const int64_t* d_len, int batch,
                               int64_t samples, int64_t* d_pred, int32_t* d_ids, int32_t* d_id_len,
                               float* d_logp, float* d_enc_len, void* d_workspace, size_t workspace_bytes,
                               vasr_stream stream);

/* The same call on int16 PCM as it sits in a wav file (AudioSegment._convert_samples_to_float32, parts/segment.py:61-74:
 * samples.astype('float32') * 2^-15): the front end's staging load converts and scales each sample as it reads it (SURVEY
 * section 8 f1) -- an exact conversion times an exact power of two, so every output equals, bit for bit, what
 * vasr_pcm16_to_f32 followed by vasr_transcribe_greedy_f32 returns; half the bytes over PCIe and into the first kernel, one
 * launch and one [B][L] float buffer fewer.  d_pcm [B][L] int16 (rows zero padded).  (ABI 7) */
VASR_API int vasr_transcribe_greedy_pcm16(vasr_handle* h, const int16_t* d_pcm, const int64_t* d_len, int batch,
                                 int64_t samples, int64_t* d_pred, int32_t* d_ids, int32_t* d_id_len,
                                 float* d_logp, float* d_enc_len, void* d_workspace, size_t workspace_bytes,
                                 vasr_stream stream);

/* GEMM arithmetic of the 1x1 convolutions of the encoder:
 *   0            v_mfma_f32_32x32x2_f32: bit-for-bit an fp32 fmaf chain;
 *   1            every fp32 operand split exactly into three bf16 terms, six cross products per multiply on
 *                v_mfma_f32_32x32x16_bf16 with fp32 accumulation (product error < one fp32 rounding, measured
 *                error against fp64 not larger than mode 0's; 2.67x less matrix time);
 *   2 (opt-in)   REDUCED precision: only the two upper bf16 terms of each operand (16 significant bits) and the
 *                three largest cross products -- half the MFMA work of mode 1 (QuartzNet15x5, B = 64: 5.5 instead of
 *                7.4 ms per batch), log-prob error against the reference goldens 5-10x mode 1's (still inside the
 *                2e-3 tolerance there, identical predictions), more accurate than the TF32 convolutions PyTorch
 *                runs by default on the GPUs the reference targets.  Never the default, never the headline number.
 *   3 (default)  every fp32 operand scaled by a power of two (exact) and split into TWO fp16 terms (22 significant
 *                bits; the scale comes from the maximum |x| of the utterance, which the kernel that produced the tensor
 *                publishes, so a row's result does not depend on the rest of its batch), the three largest cross
 *                products on v_mfma_f32_32x32x16_f16 with fp32 accumulation: half the matrix work of mode 1.  Its
 *                per-product error is larger than mode 1's (<= 3 * 2^-22) but it rounds the accumulator half as often;
 *                measured against fp64 it is not less accurate than modes 0 and 1 (tests/test_gpu_parity.py::
 *                test_split_gemms_are_as_accurate_as_fp32_mfma) and all reference fixtures hold with the same
 *                tolerances.  A GEMM whose input has no published maxima (port tensors of the per-module entry points,
 *                the CTC head) runs as mode 1.  In this mode the 256-channel separable sub-blocks (depthwise K = 33 / 39 +
 *                1x1 conv + BN + residual + ReLU) run as ONE fused kernel when the batch's 128-frame tiles fill the chip
 *                (encoder_fused.hip; its operand scale comes from a bound -- max |x| times the layer's largest tap sum --
 *                instead of the measured maximum, same tolerances).  Whether a sub-block is fused depends on the batch
 *                shape, and the two forms round differently: bit-identical rows across batch compositions are promised by
 *                vasr_set_row_independent (which never fuses), not by the default mode.
 * Layers whose shape the split kernel does not cover keep mode 0.
 * The environment variable VASR_GEMM=fp32 / bf16x3 / bf16x2 / f16x2 sets the initial mode of new handles. */
VASR_API int vasr_set_gemm_mode(vasr_handle* h, int mode);
VASR_API int vasr_get_gemm_mode(const vasr_handle* h);

/* ---- audio ingest (callers of the path: infer.py:200 librosa.load(sr=16000); parts/segment.py:19-32,61-74) ---- */
/* int16 PCM -> float32 scaled by 2^-15 (AudioSegment._convert_samples_to_float32). n = total samples. */
VASR_API int vasr_pcm16_to_f32(const int16_t* d_pcm, int64_t n, float* d_out, vasr_stream stream);
/* Band-limited sample-rate conversion of a zero-padded batch (interpolated windowed-sinc, the scheme of resampy's
 * kaiser_best that librosa.load uses by default; third-party => parity unpinned).  d_table = [nwin][2] floats
 * (window value, delta to the next entry) with num_table entries per zero crossing, built on the host
 * (viet-asr_amd/audio.py::sinc_table); ratio = sr_out / sr_in.  Lengths as librosa.load -> librosa.resample(fix=True)
 * produces them: resampy computes int(d_len_in[b] * ratio) samples, librosa pads (with zeros) to
 * d_len_out[b] = ceil(d_len_in[b] * ratio), both products in double (11 025 -> 16 000 Hz: 5 000 samples give 7 256
 * computed samples and a length of 7 257; the reference's 8 -> 16 kHz: 2 n either way).  Rows of d_out are
 * zero from int(d_len_in[b] * ratio) on.  ld_out >= ceil(ld_in * ratio).  (ABI 6; ABI 5 reported int(len * ratio).) */
VASR_API int vasr_resample_f32(const float* d_in, int64_t ld_in, const int64_t* d_len_in, int batch, const float* d_table,
                      int nwin, int num_table, double ratio, float* d_out, int64_t ld_out, int64_t* d_len_out,
                      vasr_stream stream);

/* The fused call can cut the batch into `slices` contiguous parts (1..4; default 1 = off, because on MI355X it
 * measured slower: 11.7 -> 13.5 ms at 2 slices) and run each on its own
 * internal HIP stream, forked from / joined to `stream` with events: one part's HBM-bound kernels (depthwise,
 * GEMM epilogue stores) then overlap another part's MFMA-bound GEMM main loops.  In row-independent mode results do not depend
 * on it; in the default mode every slice is a batch of its own for the shape-dependent choice between the fused and the
 * two-kernel form of a 256-channel sub-block (same tolerance, different rounding: vasr_set_gemm_mode, mode 3). */
VASR_API int vasr_set_slices(vasr_handle* h, int slices);

/* Row-independent batching (net-new; default off = the reference's batched semantics).  The reference's results
 * depend on the padded batch a signal sits in: torch.stft reflects at the end of the PADDED row (parts/features.py:
 * 181-188) and the greedy decoder collapses the padded frames too (helpers.py:7-33, quirk Q4), which is
 * why it serves one utterance per call (app.py:66-67).  With on != 0, vasr_melspec_f32 and vasr_transcribe_greedy_f32
 * treat row b as if it were alone: reflect padding at length[b], ids / id_len collapsed over the 1 + length[b] / hop mel
 * frames (taken through the conv chain) an unbatched call would have produced.  Everything in between is already
 * row-local (masks at the row's length, eval-mode BN), so ids / id_len equal those of batch-1 calls bit for bit
 * whatever the other rows are -- also in ragged batches in the fp16-split arithmetic: the encoder output's maxima (the
 * CTC head's operand scale) are taken over the row's own frames and the head reads zeros behind them.  Every length[b]
 * must exceed n_fft / 2 (an unbatched torch.stft refuses shorter input); pred / logp keep their [B, T'] shapes, frames
 * past a row's own count are unspecified. */
VASR_API int vasr_set_row_independent(vasr_handle* h, int on);
/* Compute units that another kernel of the caller's keeps busy while this handle's kernels run -- e.g. the beam search of
 * the previous batch on a side stream, one workgroup per utterance (engine.forward_beam).  The GEMM tile choice then
 * fills whole rounds of the REMAINING units: with 64 of 256 taken, 512 x 128 workgroups (one per CU) would need two
 * rounds, the second a third full; 256 x 64 ones quantise four times finer.  0 (default) = the whole device.
 * RESULTS DO NOT DEPEND ON THE HINT: it selects only between forms that give the same bits (GEMM tile shapes, the fused kernel's
 * tile width); whether a sub-block runs fused follows the batch shape alone (round 6: the hint follows a concurrent kernel's
 * progress, and with it in that decision the log-probs of a batch depended on timing). */
VASR_API int vasr_set_busy_cus(vasr_handle* h, int cus);

/* ---- beam search (+ n-gram LM) ------------------------------------------------------------ */
/* BeamSearchDecoderWithLM.forward (beam_search_decoder.py:95-102 -> pyctcdecode, third-party: parity unpinned,
 * algorithm restated in oracle/beam_oracle.py).  Unlike the reference any batch size is accepted.
 *   d_logp [B][T'][V+1] f32 log-probabilities, blank = V (last class); space_id = index of ' ' in the labels
 *   -> d_ids [B][T'] i32 label ids of the best hypothesis (words separated by space_id), d_id_len [B] i32 (-1: the
 *      four-wavefront kernel's merge cells overflowed -- a prefix is reached by at most four pairs, so this cannot happen; it is
 *      reported instead of a wrong answer),
 *      d_score [B] f32 combined (acoustic + LM) natural-log score of that hypothesis.
 * beam_width <= 128, V+1 <= 128.  token_min_logp / beam_prune_logp: pyctcdecode defaults are -5 / -10. */
typedef struct vasr_lm vasr_lm;
VASR_API size_t vasr_beam_workspace_bytes(int batch, int64_t frames);
/* Compute units a search of `batch` utterances occupies while it runs: what a caller that overlaps the search with the next
 * acoustic pass hands to vasr_set_busy_cus().  Up to 64 utterances an utterance is searched by four wavefronts of a compute
 * unit of its own (`batch` units, for the shortest time); beyond that by one wavefront, four utterances per workgroup = per
 * compute unit (ceil(batch / 4) units).  Both forms give the same bits. */
VASR_API int vasr_beam_workgroups(int batch);
VASR_API int vasr_beam_search_f32(const float* d_logp, int batch, int64_t frames, int num_classes, int space_id,
                         int beam_width, float token_min_logp, float beam_prune_logp, const vasr_lm* lm,
                         int32_t* d_ids, int32_t* d_id_len, float* d_score, void* d_workspace,
                         size_t workspace_bytes, vasr_stream stream);
/* Same with a frame count per row: d_row_frames [B] i32 (device), row b is searched over its first
 * min(d_row_frames[b], frames) frames -- for batches of utterances of different lengths whose rows must come out as
 * batch-1 calls would (vasr_set_row_independent); NULL = all frames for every row (the call above). */
VASR_API int vasr_beam_search_rows_f32(const float* d_logp, const int32_t* d_row_frames, int batch, int64_t frames,
                              int num_classes, int space_id, int beam_width, float token_min_logp,
                              float beam_prune_logp, const vasr_lm* lm, int32_t* d_ids, int32_t* d_id_len,
                              float* d_score, void* d_workspace, size_t workspace_bytes, vasr_stream stream);
/* Back-off n-gram model as two open-addressing hash tables of 16-BYTE entries (one load returns key and value), both with
 * power-of-two capacity 2^lg >= 16, linear probing from the home slot ((uint32)(key ^ key >> 32) * 0x9E3779B1) >> (32 - lg),
 * key 0 = empty slot, stored keys have bit 0 set:
 *   h_vocab [vcap] {uint64 key, int32 word id, uint32 flags}: key = the word's label ids folded with vasr_beam_hash_step from
 *           vasr_beam_hash_init(), first character first; flags bit 0 = the word is in pyctcdecode's unigram set (below);
 *   h_ngram [ncap] {uint64 key, float log10 p, float log10 back-off}: the key of (w_1 .. w_n) folds the word ids from the
 *           LAST word backwards, hash_step(... hash_step(hash_step(init, w_n), w_{n-1}) ..., w_1) -- the keys of every suffix
 *           of a history then come out of one chain, and the kernel requests the whole back-off walk in one trip to memory.
 *   h_trie  [trie_buckets] buckets of TWO uint64 keys (16 bytes), trie_buckets a power of two >= 16, or NULL.
 *           pyctcdecode's build_ctcdecoder(labels, kenlm_model_path, alpha, beta) -- the reference's call,
 *           beam_search_decoder.py:82-87 -- behaves in one of two ways depending on the path's SUFFIX: for "*.arpa" it reads
 *           the file's unigrams (1-gram lines with a back-off field), keeps those the model knows (unigram_set) and builds a
 *           character trie of them: a partial word that is a prefix of a set member carries NO out-of-vocabulary penalty,
 *           and a committed word outside the set gets the unk offset even if the model knows it.  For any other suffix (the
 *           reference's own `3-gram-lm.binary`) there is no set: every partial word is penalised.
 *           h_trie = the trie's nodes: key = (label ids of a non-empty PREFIX of a set member folded like a word's) | 1, in its
 *           home bucket ((uint32)(key ^ key >> 32) * 0x9E3779B1) >> (32 - lg buckets) or the next one with a free cell (0);
 *           NULL = the no-unigram behaviour (also what an EMPTY unigram set amounts to).
 * Host arrays are copied to the device.  alpha/beta/unk_offset as in pyctcdecode's LanguageModel.  (ABI 6; ABI 5 had no
 * trie; ABI 4 took four parallel arrays with odd capacities and `key % cap`.) */
VASR_API int vasr_lm_create(const void* h_vocab, int vcap, const void* h_ngram, int ncap, const void* h_trie, int trie_buckets,
                   int order, int bos_id, int eos_id, int unk_id, float alpha, float beta, float unk_offset, vasr_lm** out);
VASR_API void vasr_lm_destroy(vasr_lm* lm);
VASR_API uint64_t vasr_beam_hash_init(void);
VASR_API uint64_t vasr_beam_hash_step(uint64_t h, uint64_t v);

/* ---- introspection -------------------------------------------------------------------- */
VASR_API const char* vasr_last_error(void);
VASR_API const char* vasr_version(void);
VASR_API int vasr_abi_version(void);
/* Algorithmic work of one call at (batch, samples): flops of the 1x1-conv GEMMs, flops and
 * minimum HBM bytes (read input + write output + weights, fp32) of the depthwise layers.
 * out[0]=pointwise_flops out[1]=depthwise_flops out[2]=depthwise_bytes out[3]=decoder_flops
 * out[4]=frontend_flops (2.5 N log2 N per frame + mel) */
VASR_API int vasr_algorithmic_work(const vasr_handle* h, int batch, int64_t samples, double out[5]);
/* Per-kernel-class timing with HIP events recorded on the launch stream (used by bench.py for the
 * roofline figures).  Between begin and end every launch of vasr_transcribe_greedy_f32 /
 * vasr_encoder_f32 / vasr_decoder_* is bracketed by an event pair; end() synchronises on the events
 * and returns the summed milliseconds and launch counts per class:
 *   [0] front end (seq_len + STFT/mel + CMVN)  [1] depthwise convs  [2] pointwise GEMMs
 *   [3] CTC head (decoder GEMM + log-softmax/argmax + collapse)  [4] fused depthwise + pointwise sub-blocks
 * flops / bytes (optional, may be NULL): the algorithmic work of the launches of each class that actually ran --
 * 2 M N K of every GEMM (class 2 and 4), HBM bytes read + written by every depthwise (1) and fused (4) layer, bytes STORED by
 * every plain GEMM (2: its store-only epilogue is the part of its time the matrix pipe does not bound). */
VASR_API int vasr_profile_begin(vasr_handle* h);
VASR_API int vasr_profile_end(vasr_handle* h, double ms[5], int64_t launches[5], double flops[5], double bytes[5]);
/* Row pitch of the library's padded activation buffers: frames rounded up to the 128-frame tile. */
VASR_API int64_t vasr_padded_frames(int64_t frames);

#ifdef __cplusplus
}
#endif
/* Measurement / development entry points (isolated layers, weight packers, bracket overhead) live in vasr_devtools.h and
 * are exported only by libvasr_hip_dev.so (-DVASR_DEVTOOLS); the product library has none of them and reads no
 * VASR_* tuning variable from the environment except VASR_GEMM and VASR_SLICES (the documented modes above). */
#endif /* VASR_H_ */
