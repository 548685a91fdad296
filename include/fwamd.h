/*
 * fwamd.h — C ABI of the MI355X-native Whisper engine (libfwamd.so).
 *
 * This is the drop-in boundary for the hot path of SYSTRAN/faster-whisper:
 * everything the reference reaches through `ctranslate2.models.Whisper`
 * (reference call sites, all in faster_whisper/transcribe.py):
 *
 *   constructor        transcribe.py:689-698   -> fw_model_create / fw_model_free
 *   .is_multilingual   transcribe.py:379,472   -> fw_model_info
 *   .n_mels            transcribe.py:484       -> fw_model_info
 *   .device/.device_index transcribe.py:1394   -> fw_model_info
 *   StorageView.from_array transcribe.py:1875  -> host float* handed to fw_encode
 *   .encode            transcribe.py:1400      -> fw_encode
 *   .generate          transcribe.py:222-236, 1446-1459 -> fw_generate
 *   .detect_language   transcribe.py:215,1193,1823      -> fw_detect_language
 *   .align             transcribe.py:1709-1715          -> fw_align
 *
 * plus the log-mel front end the reference runs in numpy on the host
 * (faster_whisper/feature_extractor.py:198-230, called at
 * transcribe.py:463-467 and padded by audio.py:111-123) -> fw_logmel, and a
 * fused PCM -> mel -> encoder entry (fw_encode_pcm) that keeps the features
 * resident in HBM.
 *
 * Conventions
 *  - plain C types only; no torch / HIP types cross this boundary.
 *  - every entry point returns 0 on success, a negative FW_E* code on error;
 *    fw_last_error() returns a thread-local message for the last failure.
 *    The Python shim maps FW_EINVAL -> ValueError, everything else ->
 *    RuntimeError (CTranslate2 surfaces std::invalid_argument / runtime_error
 *    the same way).
 *  - host pointers unless the name says `_dev`.
 *  - the library never falls back to a CPU implementation: without a HIP
 *    device fw_model_create fails with FW_ENODEV.
 */
#ifndef FWAMD_H
#define FWAMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FW_OK 0
#define FW_EINVAL (-1)   /* bad argument / shape (-> ValueError)      */
#define FW_ENODEV (-2)   /* no HIP device / HIP runtime failure       */
#define FW_ENOMEM (-3)   /* device or host allocation failed          */
#define FW_ERUNTIME (-4) /* kernel launch / internal error            */
#define FW_ENOSPC (-5)   /* a caller-provided output buffer is too small (fw_flac_decode): retry with a larger one */

/* 2: fw_model_set_encoder_cus / fw_model_encoder_cus removed, fw_model_set_merge_wait, fw_model_set_decode_lanes and
 *    fw_model_run_capacity added, test / bench hooks moved to fwamd_test.h; the cross-attention cache of a decode group
 *    is one pool shared by its lanes (fw_model_decode_batch = chunks of the pool, fw_model_run_capacity = chunks of one run) */
#define FW_ABI_VERSION 2

/* compute types (the reference passes the CTranslate2 strings, transcribe.py:626) */
#define FW_COMPUTE_FLOAT16 0      /* "float16" / "default" on GPU */
#define FW_COMPUTE_INT8_FLOAT16 1 /* "int8_float16": int8 weights + dynamic int8 activations, fp16 elsewhere */

/* weight element types accepted by fw_model_create */
#define FW_DT_F32 0
#define FW_DT_F16 1

#define FW_MAX_ALIGN_HEADS 64

/* Model geometry + vocabulary layout (what CTranslate2 reads from model.bin /
 * config.json). */
typedef struct fw_config {
  int32_t n_mels;         /* 80 or 128 */
  int32_t n_audio_ctx;    /* 1500 */
  int32_t d_model;        /* 384 .. 1280, multiple of 128 */
  int32_t n_heads;        /* d_model / 64 */
  int32_t n_enc_layers;
  int32_t n_dec_layers;
  int32_t n_vocab;        /* 51864 / 51865 / 51866 */
  int32_t n_text_ctx;     /* 448 */
  int32_t is_multilingual;
  /* special token ids */
  int32_t tok_eot, tok_sot, tok_lang_begin, n_langs;
  int32_t tok_translate, tok_transcribe, tok_sot_lm, tok_sot_prev;
  int32_t tok_no_speech, tok_no_timestamps, tok_timestamp_begin;
  /* suppress_blank id set (config.json: suppress_ids_begin): " " token and eot */
  int32_t n_suppress_begin;
  int32_t suppress_begin[8];
  /* alignment heads (config.json: alignment_heads) as (layer, head) pairs;
   * n_align_heads == 0 -> all heads of the upper half of the decoder */
  int32_t n_align_heads;
  int32_t align_heads[2 * FW_MAX_ALIGN_HEADS];
} fw_config;

/* One named weight tensor (names: see faster_whisper_amd/weights.py). */
typedef struct fw_weight {
  const char* name;
  const void* data; /* host pointer, C-contiguous */
  int32_t dtype;    /* FW_DT_* */
  int32_t ndim;
  int64_t dims[4];
} fw_weight;

typedef struct fw_model fw_model;   /* opaque: weights + workspaces on one GPU */
typedef struct fw_tensor fw_tensor; /* opaque: device-resident encoder output [B, n_audio_ctx, d_model] fp16 */

/* generate() options — the kwargs of ctranslate2.models.Whisper.generate that
 * faster-whisper passes (transcribe.py:222-236 and :1433-1459). */
typedef struct fw_gen_opts {
  int32_t beam_size;            /* 1 = greedy */
  float patience;               /* default 1 */
  int32_t num_hypotheses;       /* default 1 */
  float length_penalty;         /* default 1 */
  float repetition_penalty;     /* default 1 (disabled) */
  int32_t no_repeat_ngram_size; /* default 0 (disabled) */
  int32_t max_length;           /* default 448; counts the prompt */
  int32_t return_scores;
  int32_t return_no_speech_prob;
  int32_t max_initial_timestamp_index; /* default 50 */
  int32_t suppress_blank;       /* default 1 */
  const int32_t* suppress_tokens; /* may be NULL; -1 entries are ignored */
  int32_t n_suppress_tokens;
  int32_t sampling_topk;        /* default 1; 0 = full distribution */
  float sampling_temperature;   /* default 1; used only when beam_size==1 and (topk != 1) */
  uint64_t seed;                /* sampling seed (CTranslate2 uses a global seed) */
  /* benchmark-only control (not part of the reference API): when > 0, EOT is
   * suppressed for the first `min_new_tokens` generated tokens, so synthetic
   * weights decode a fixed, input-independent length. */
  int32_t min_new_tokens;
} fw_gen_opts;

const char* fw_last_error(void);
int32_t fw_abi_version(void);

/* Number of visible HIP devices (0 when none / runtime missing). */
int32_t fw_device_count(void);

/* ---- model -------------------------------------------------------------- */
/* Replaces ctranslate2.models.Whisper(...) (transcribe.py:689-698) for ONE
 * device index; the Python shim keeps one fw_model per entry of device_index.
 * Weights are copied to HBM (converted to fp16, or quantised per output row
 * to int8 for FW_COMPUTE_INT8_FLOAT16); the host arrays may be freed after
 * the call returns. max_batch bounds B in every later call (workspaces are
 * allocated once, here). */
int32_t fw_model_create(const fw_config* cfg, const fw_weight* weights, int32_t n_weights,
                        int32_t compute_type, int32_t device_index, int32_t max_batch,
                        int32_t max_beam, fw_model** out);
void fw_model_free(fw_model* m);
int32_t fw_model_info(const fw_model* m, fw_config* cfg_out, int32_t* compute_type,
                      int32_t* device_index, int32_t* max_batch, int32_t* max_beam);
/* Device address / size of a model's weight blob: lets N worker replicas on one GPU share one copy
 * of the weights (CTranslate2's inter_threads / faster-whisper's num_workers, transcribe.py:654-657):
 * create the extra replicas with fw_model_create_from_blob_dev on this pointer.  fw_model_free of the owning
 * model (and of the primary of a decode group) is deferred until its last dependent has been freed. */
int32_t fw_model_blob(const fw_model* m, void** blob_dev, int64_t* blob_bytes);
/* Same model from a weight blob that is ALREADY in HBM on `device_index`
 * (the RCCL-broadcast path: rank 0 packs, ranks receive into device memory).
 * blob layout: faster_whisper_amd/weights.py::pack_blob. */
int32_t fw_model_create_from_blob_dev(const fw_config* cfg, const void* blob_dev, int64_t blob_bytes,
                                      int32_t compute_type, int32_t device_index,
                                      int32_t max_batch, int32_t max_beam, fw_model** out);

/* Decode groups — what CTranslate2's replica pool (inter_threads; transcribe.py:645-657, :689-698) becomes on a
 * GPU with 288 GB of HBM.  A decode step streams every decoder weight once whatever the number of rows, so the
 * worker replicas of a device share ONE decode workspace instead of decoding side by side:
 *   fw_model_set_decode_batch(primary, n_workers * max_batch)  sizes the group for that many chunks in flight: the
 *       cross-attention K / V^T pool holds them (rounded down to whole encoder batches and to what fits in 70 % of the free
 *       HBM; fw_model_decode_batch reads the pool's chunk count back), ONE copy whatever the number of lanes: an encoder
 *       output's block of the pool is handed to whichever lane decodes it.  A decode RUN holds at most
 *       fw_model_run_capacity chunks: 2048 rows with one lane, 1600 rows per lane with two (from four encoder batches on
 *       the group decodes on TWO lanes — two run workspaces, two streams: two decode runs are in flight at once, the
 *       HBM-bound cross-attention of one beside the linears of the other).  The self-attention cache of a lane is rows x
 *       positions and every run lays it out for its own max_length, so a run of calls with a short max_length holds more
 *       chunks than one that asks for the whole text context (positions per row are lowered, down to 160, before the
 *       pool is).  Setup-time call: FW_EINVAL while the group has decode runs queued or in flight; fw_generate calls that
 *       arrive during the rebuild wait for it;
 *   fw_model_join_decoder(worker, primary)  frees the worker's own decode workspace and routes its fw_generate /
 *       fw_detect_language / fw_align calls to the primary's.
 * fw_generate calls with identical options that arrive from different host threads while a decode run is in
 * progress are merged into the next run (up to decode_batch chunks); each caller gets exactly the result it would
 * get alone.  Encoders keep running per worker on their own streams and overlap with the running decode. */
int32_t fw_model_set_decode_batch(fw_model* m, int32_t decode_batch);
int32_t fw_model_decode_batch(const fw_model* m);
/* chunks one decode run of the group can hold (<= fw_model_decode_batch) */
int32_t fw_model_run_capacity(const fw_model* m);
int32_t fw_model_join_decoder(fw_model* worker, fw_model* primary);
/* How long, and for how much, the leader of a decode run waits for the requests of workers that are still encoding
 * (latency of the waiting call against rows per run).  wait_ms: -1 = 2.5 measured encoder passes after the last arrival,
 * at most 250 ms (default); 0 = never wait (every call starts its run at once: lowest latency); n > 0 = n milliseconds.
 * fill_percent: the leader stops waiting once that share of a run's chunk capacity is queued (default 90: large runs
 * amortise the decoder weights and launches over more rows and take the GEMM-shaped decoder linears; measured 2 917x
 * at 50, 2 965x at 75, 2 975x at 95; smaller = lower latency).  It never waits when no member encode is in flight.  A group with NO run in progress and two
 * lanes leads earlier: it splits the work it knows of (queued requests + one per worker inside an encode call) evenly over
 * the runs that work needs — 20 batches in flight become two runs of 10 instead of one of 18 after 18 serial encoder
 * passes and leftovers (FWAMD_IDLE_BALANCE=0 restores the plain rule). */
int32_t fw_model_set_merge_wait(fw_model* m, int32_t wait_ms, int32_t fill_percent);
/* Decode runs the group may have in flight: 2 (default, when the group has two lanes) or 1 (one run at a time: every
 * kernel of a run has the chip to itself — what per-kernel timing wants; takes effect with the next run).  A group built
 * with FWAMD_DECODE_LANES = 3 / 4 in the environment (measurement knob) has that many lanes and takes 1 .. that many. */
int32_t fw_model_set_decode_lanes(fw_model* m, int32_t lanes);
/* counters of the decode group `m` belongs to: decode runs, fw_generate calls served, chunks decoded, chunks of
 * the largest run (any pointer may be NULL) */
int32_t fw_model_decode_stats(const fw_model* m, int64_t* runs, int64_t* requests, int64_t* chunks,
                              int32_t* max_run_chunks);

/* Host image of the device weight blob (what rank 0 broadcasts over RCCL/xGMI at load):
 * fw_pack_blob_size packs and returns a handle + byte size, fw_pack_blob_copy copies the
 * image into caller memory, fw_pack_blob_free releases the handle. */
int32_t fw_pack_blob_size(const fw_config* cfg, const fw_weight* weights, int32_t n_weights,
                          int32_t compute_type, int64_t* size_out, void** handle_out);
int32_t fw_pack_blob_copy(void* handle, void* dst, int64_t dst_bytes);
void fw_pack_blob_free(void* handle);

/* ---- log-mel front end ---------------------------------------------------
 * FeatureExtractor.__call__(chunk)[..., :-1] followed by pad_or_trim(., 3000)
 * for B ragged chunks (feature_extractor.py:198-230, transcribe.py:463-467,
 * 514-516, audio.py:111-123). pcm: the B chunks back to back; offsets[B+1]
 * sample offsets into pcm (chunk b = pcm[offsets[b] .. offsets[b+1])).
 * out: float32 [B, n_mels, 3000]. n_frames_out (may be NULL): frames that
 * carry signal per chunk (min(3000, n_b/160)). */
int32_t fw_logmel(fw_model* m, const float* pcm, const int64_t* offsets, int32_t B,
                  float* out, int32_t* n_frames_out);
/* Whole-waveform variant used by the sequential path and detect_language:
 * FeatureExtractor.__call__(waveform) without the trailing-frame drop and
 * without pad_or_trim (transcribe.py:916). out: [n_mels, n_samples/160 + 1]. */
int32_t fw_logmel_full(fw_model* m, const float* pcm, int64_t n_samples, float* out, int64_t out_frames);

/* ---- encoder -------------------------------------------------------------
 * ctranslate2.models.Whisper.encode (transcribe.py:1400): features float32
 * [B, n_mels, 3000] on the host -> device-resident encoder output handle. */
int32_t fw_encode(fw_model* m, const float* features, int32_t B, fw_tensor** out);
/* Fused resident path: ragged PCM -> log-mel -> encoder without the features
 * ever leaving HBM (same numerics as fw_logmel + fw_encode). */
int32_t fw_encode_pcm(fw_model* m, const float* pcm, const int64_t* offsets, int32_t B, fw_tensor** out);
/* As fw_encode_pcm, but pcm_dev already resides in HBM (bench.py: inputs are
 * resident when the timed region starts). */
int32_t fw_encode_pcm_dev(fw_model* m, const float* pcm_dev, const int64_t* offsets, int32_t B, fw_tensor** out);
int32_t fw_tensor_shape(const fw_tensor* t, int32_t* B, int32_t* T, int32_t* D);
/* encoder output as float32 [B, T, D] on the host (StorageView -> numpy; to_cpu=True) */
int32_t fw_tensor_to_host(fw_model* m, const fw_tensor* t, float* out);
/* upload a host float32 [B, T, D] encoder output (the to_cpu round trip, transcribe.py:1392-1394) */
int32_t fw_tensor_from_host(fw_model* m, const float* data, int32_t B, fw_tensor** out);
void fw_tensor_free(fw_tensor* t);

/* ---- generate ------------------------------------------------------------
 * ctranslate2.models.Whisper.generate. prompts: B prompts back to back,
 * prompt_offsets[B+1]. Outputs, per chunk b and hypothesis h < num_hypotheses:
 *   out_ids    int32 [B, num_hypotheses, max_length]  generated ids (no prompt, no eot)
 *   out_lens   int32 [B, num_hypotheses]
 *   out_scores float [B, num_hypotheses]  cum_logprob / len^length_penalty (0 if !return_scores)
 *   out_no_speech float [B]               (0 if !return_no_speech_prob)
 */
int32_t fw_generate(fw_model* m, const fw_tensor* enc, const int32_t* prompts,
                    const int32_t* prompt_offsets, int32_t B, const fw_gen_opts* opts,
                    int32_t* out_ids, int32_t* out_lens, float* out_scores, float* out_no_speech);

/* ---- detect_language -------------------------------------------------------
 * ctranslate2.models.Whisper.detect_language: one decoder step on [sot];
 * softmax restricted to the language ids. out_lang_ids / out_probs:
 * [B, n_langs], sorted by descending probability. */
int32_t fw_detect_language(fw_model* m, const fw_tensor* enc, int32_t B,
                           int32_t* out_lang_ids, float* out_probs);

/* ---- align -----------------------------------------------------------------
 * ctranslate2.models.Whisper.align (transcribe.py:1709-1715).
 * start_seq[n_start]: sot sequence; text tokens ragged via text_offsets[B+1];
 * num_frames[B] (mel frames, the reference passes segment_size).
 * Outputs: out_pairs int32 [B, max_pairs, 2] (text_idx, time_idx) with
 * out_n_pairs[B]; out_probs float ragged like text tokens (same offsets).
 * text_idx runs over n_text + 1 rows: row 0 is the <|notimestamps|> position
 * (its attention times the first text token), row n_text the last text token;
 * the reference looks the path's token jumps up with word boundaries 0..n_text
 * (transcribe.py:1741-1745).  time_idx < num_frames / 2.
 * max_pairs >= n_text + 1 + num_frames / 2. */
int32_t fw_align(fw_model* m, const fw_tensor* enc, const int32_t* start_seq, int32_t n_start,
                 const int32_t* text_tokens, const int32_t* text_offsets, const int32_t* num_frames,
                 int32_t B, int32_t median_filter_width, int32_t max_pairs,
                 int32_t* out_pairs, int32_t* out_n_pairs, float* out_probs);

/* ---- measurement hooks (bench.py) ------------------------------------------
 * Per-kernel-family GPU time accumulated with HIP events on the engine's own
 * stream while profiling is enabled. names: fw_prof_name(i), i < fw_prof_count(). */
void fw_prof_enable(fw_model* m, int32_t on);
void fw_prof_reset(fw_model* m);
int32_t fw_prof_count(void);
const char* fw_prof_name(int32_t i);
/* total ms, launches, algorithmic flops, algorithmic bytes for family i */
int32_t fw_prof_get(fw_model* m, int32_t i, double* ms, int64_t* launches, double* flops, double* bytes);
/* block until all work queued on the model's streams has finished */
int32_t fw_synchronize(fw_model* m);
/* raw device allocation helpers so bench.py can stage PCM in HBM without torch */
int32_t fw_dev_alloc(fw_model* m, int64_t bytes, void** out_dev);
int32_t fw_dev_free(fw_model* m, void* dev);
int32_t fw_dev_upload(fw_model* m, void* dst_dev, const void* src_host, int64_t bytes);

/* ---- Silero VAD network (host) ----------------------------------------------
 * Replaces the reference's SileroVADModel (faster_whisper/vad.py:295-351: onnxruntime on the CPU, one thread,
 * asset silero_vad_v6.onnx).  Host C++: needs no GPU.  The weights are the initializers of that ONNX file, passed
 * as plain float32 arrays (faster_whisper_amd/onnx_lite.py reads them); the architecture is fixed to Silero v6:
 *   stft_basis [258][256]; conv_w {[128][129][3], [64][128][3], [64][64][3], [128][64][3]} + conv_b;
 *   lstm_w / lstm_r [512][128] in ONNX gate order i,o,f,c; lstm_b [1024] = Wb | Rb; dec_w [128], dec_b.
 * fw_vad_forward: windows [n][576] float32 (64 samples of context + 512 new samples each, exactly what
 * SileroVADModel.__call__ feeds the session, vad.py:318-336); h, c [128] LSTM state, updated in place (the
 * windows are the LSTM's sequence; the reference carries h / c across its batches of 10 000 windows);
 * probs [n] speech probability per window.  n_threads <= 0: all host cores. */
typedef struct fw_vad fw_vad;
typedef struct fw_vad_weights {
  const float* stft_basis;
  const float* conv_w[4];
  const float* conv_b[4];
  const float* lstm_w;
  const float* lstm_r;
  const float* lstm_b;
  const float* dec_w;
  float dec_b;
} fw_vad_weights;
int32_t fw_vad_create(const fw_vad_weights* w, fw_vad** out);
int32_t fw_vad_forward(fw_vad* v, const float* windows, int64_t n, int32_t n_threads, float* h, float* c,
                       float* probs);
void fw_vad_free(fw_vad* v);
/* Same computation on HIP device `device_index` (csrc/vad.hip: one workgroup per window for the front end, one
 * persistent workgroup for the LSTM recurrence); host pointers in and out.  Checked on hardware against the host
 * path and against the numpy restatement of the ONNX graph (tests/test_gpu_vad.py); the Python front uses the host
 * path by default (the reference runs the VAD on the CPU, vad.py:295-351). */
int32_t fw_vad_forward_dev(fw_vad* v, int32_t device_index, const float* windows, int64_t n, float* h, float* c,
                           float* probs);
/* The same network over a whole RECORDING: `audio` = n_samples float32 samples in host memory, n_samples a multiple of 512
 * (the reference pads with 1..512 zeros, vad.py:79-81); the 576-sample rows [64 context | 512 window] the network takes are
 * formed on the device exactly as SileroVADModel.__call__ forms them on the host (faster_whisper/vad.py:318-336: context
 * = tail of the previous window, zeros for the first; the last 64 samples of the last window zeroed), so no [n][576] copy
 * of the recording is built or transferred.  probs: n_samples / 512 values. */
int32_t fw_vad_forward_audio_dev(fw_vad* v, int32_t device_index, const float* audio, int64_t n_samples, float* h, float* c,
                                 float* probs);

/* ---- audio front: native FLAC decoding --------------------------------------------
 * The reference decodes every container through PyAV / FFmpeg (faster_whisper/audio.py:19-76); its own test asset
 * (tests/data/jfk.flac) is FLAC.  fw_flac_info reads STREAMINFO (total_samples is per channel, 0 = unknown);
 * fw_flac_decode decodes the whole stream (host code, csrc/flac_host.cpp) into interleaved int32 samples
 * out[sample][channel] (capacity_samples per channel), verifying every frame's CRC-8 / CRC-16, and reports in md5_status
 * whether the decoded PCM carries the MD5 signature the encoder stored: 1 = yes (bit-exact decode), 0 = no, -1 = nothing
 * to compare with (no signature, or a truncated stream: the whole frames present are returned).
 * FW_ENOSPC: the stream holds more than capacity_samples per channel (nothing usable was written): retry with a larger buffer. */
int32_t fw_flac_info(const uint8_t* data, int64_t n_bytes, int32_t* sample_rate, int32_t* channels,
                     int32_t* bits_per_sample, int64_t* total_samples);
int32_t fw_flac_decode(const uint8_t* data, int64_t n_bytes, int32_t* out, int64_t capacity_samples,
                       int64_t* n_decoded, int32_t* md5_status);

#ifdef __cplusplus
}
#endif
#endif /* FWAMD_H */
