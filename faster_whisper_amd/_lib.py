"""ctypes binding of libfwamd.so (include/fwamd.h).  No CPU fallback: if the library is
missing or no HIP device is visible, model construction raises."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# FWAMD_LIB: another build of the library (A/B of two builds on one GPU box: profiles/calls/r04_ab_builds.sh); the product
# loads the in-tree libfwamd.so
LIB_PATH = os.environ.get("FWAMD_LIB") or os.path.join(_HERE, "libfwamd.so")

FW_OK = 0
FW_EINVAL = -1
FW_ENODEV = -2
FW_ENOMEM = -3
FW_ERUNTIME = -4
FW_ENOSPC = -5

COMPUTE_FLOAT16 = 0
COMPUTE_INT8_FLOAT16 = 1
FW_DT_F32 = 0
FW_DT_F16 = 1
FW_MAX_ALIGN_HEADS = 64


class FwConfig(C.Structure):
    _fields_ = [
        ("n_mels", C.c_int32), ("n_audio_ctx", C.c_int32), ("d_model", C.c_int32), ("n_heads", C.c_int32),
        ("n_enc_layers", C.c_int32), ("n_dec_layers", C.c_int32), ("n_vocab", C.c_int32),
        ("n_text_ctx", C.c_int32), ("is_multilingual", C.c_int32),
        ("tok_eot", C.c_int32), ("tok_sot", C.c_int32), ("tok_lang_begin", C.c_int32), ("n_langs", C.c_int32),
        ("tok_translate", C.c_int32), ("tok_transcribe", C.c_int32), ("tok_sot_lm", C.c_int32),
        ("tok_sot_prev", C.c_int32), ("tok_no_speech", C.c_int32), ("tok_no_timestamps", C.c_int32),
        ("tok_timestamp_begin", C.c_int32),
        ("n_suppress_begin", C.c_int32), ("suppress_begin", C.c_int32 * 8),
        ("n_align_heads", C.c_int32), ("align_heads", C.c_int32 * (2 * FW_MAX_ALIGN_HEADS)),
    ]


class FwWeight(C.Structure):
    _fields_ = [("name", C.c_char_p), ("data", C.c_void_p), ("dtype", C.c_int32), ("ndim", C.c_int32),
                ("dims", C.c_int64 * 4)]


class FwGenOpts(C.Structure):
    _fields_ = [
        ("beam_size", C.c_int32), ("patience", C.c_float), ("num_hypotheses", C.c_int32),
        ("length_penalty", C.c_float), ("repetition_penalty", C.c_float), ("no_repeat_ngram_size", C.c_int32),
        ("max_length", C.c_int32), ("return_scores", C.c_int32), ("return_no_speech_prob", C.c_int32),
        ("max_initial_timestamp_index", C.c_int32), ("suppress_blank", C.c_int32),
        ("suppress_tokens", C.POINTER(C.c_int32)), ("n_suppress_tokens", C.c_int32),
        ("sampling_topk", C.c_int32), ("sampling_temperature", C.c_float), ("seed", C.c_uint64),
        ("min_new_tokens", C.c_int32),
    ]


class FwVadWeights(C.Structure):
    _fields_ = [
        ("stft_basis", C.c_void_p), ("conv_w", C.c_void_p * 4), ("conv_b", C.c_void_p * 4),
        ("lstm_w", C.c_void_p), ("lstm_r", C.c_void_p), ("lstm_b", C.c_void_p), ("dec_w", C.c_void_p),
        ("dec_b", C.c_float),
    ]


# every symbol include/fwamd.h and include/fwamd_test.h declare (tests/test_abi.py checks the .so exports them all)
SYMBOLS = [
    "fw_last_error", "fw_abi_version", "fw_device_count",
    "fw_model_create", "fw_model_free", "fw_model_info", "fw_model_blob", "fw_model_create_from_blob_dev",
    "fw_model_set_decode_batch", "fw_model_decode_batch", "fw_model_join_decoder", "fw_model_decode_stats",
    "fw_model_set_merge_wait", "fw_model_set_decode_lanes", "fw_model_run_capacity", "fw_dec_big_min_rows", "fw_dec_big_min_rows_of", "fw_test_knob",
    "fw_test_idle_lead_chunks",
    "fw_pack_blob_size", "fw_pack_blob_copy", "fw_pack_blob_free",
    "fw_logmel", "fw_logmel_full",
    "fw_encode", "fw_encode_pcm", "fw_encode_pcm_dev", "fw_tensor_shape", "fw_tensor_to_host",
    "fw_tensor_from_host", "fw_tensor_free",
    "fw_generate", "fw_detect_language", "fw_align",
    "fw_prof_enable", "fw_prof_reset", "fw_prof_count", "fw_prof_name", "fw_prof_get", "fw_synchronize",
    "fw_dev_alloc", "fw_dev_free", "fw_dev_upload",
    "fw_test_gemm", "fw_test_layernorm", "fw_test_attention", "fw_test_dec_linear", "fw_test_dec_logits", "fw_test_logits_rules", "fw_bench_gemm", "fw_bench_dec_linear", "fw_bench_attention",
    "fw_vad_create", "fw_vad_forward", "fw_vad_free", "fw_vad_forward_dev", "fw_vad_forward_audio_dev",
    "fw_flac_info", "fw_flac_decode",
]

_lib = None


class FwError(RuntimeError):
    pass


def load():
    """Load libfwamd.so (built by __graft_entry__.build()).  Raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FwError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). faster_whisper_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    vp, i32, i64, f32p = C.c_void_p, C.c_int32, C.c_int64, C.POINTER(C.c_float)
    i32p, i64p = C.POINTER(C.c_int32), C.POINTER(C.c_int64)
    lib.fw_last_error.restype = C.c_char_p
    lib.fw_abi_version.restype = i32
    lib.fw_device_count.restype = i32
    lib.fw_model_create.argtypes = [C.POINTER(FwConfig), C.POINTER(FwWeight), i32, i32, i32, i32, i32, C.POINTER(vp)]
    lib.fw_model_free.argtypes = [vp]
    lib.fw_model_free.restype = None
    lib.fw_model_info.argtypes = [vp, C.POINTER(FwConfig), i32p, i32p, i32p, i32p]
    lib.fw_model_blob.argtypes = [vp, C.POINTER(vp), i64p]
    lib.fw_model_create_from_blob_dev.argtypes = [C.POINTER(FwConfig), vp, i64, i32, i32, i32, i32, C.POINTER(vp)]
    lib.fw_model_set_decode_batch.argtypes = [vp, i32]
    lib.fw_model_decode_batch.argtypes = [vp]
    lib.fw_model_decode_batch.restype = i32
    lib.fw_model_run_capacity.argtypes = [vp]
    lib.fw_model_run_capacity.restype = i32
    lib.fw_dec_big_min_rows.restype = i32
    if hasattr(lib, "fw_dec_big_min_rows_of"):   # (absent from an older build loaded through FWAMD_LIB)
        lib.fw_dec_big_min_rows_of.restype = i32
        lib.fw_dec_big_min_rows_of.argtypes = [i32, i32]
    if hasattr(lib, "fw_flac_info"):
        lib.fw_flac_info.argtypes = [vp, i64, i32p, i32p, i32p, i64p]
        lib.fw_flac_decode.argtypes = [vp, i64, vp, i64, i64p, i32p]
    if hasattr(lib, "fw_test_knob"):          # (absent from the older build an A/B loads through FWAMD_LIB)
        lib.fw_test_knob.argtypes = [i32, i32]
    if hasattr(lib, "fw_test_idle_lead_chunks"):
        lib.fw_test_idle_lead_chunks.argtypes = [i64, i32, i32, i64, i32]
        lib.fw_test_idle_lead_chunks.restype = i64
    lib.fw_model_join_decoder.argtypes = [vp, vp]
    lib.fw_model_set_merge_wait.argtypes = [vp, i32, i32]
    lib.fw_model_set_decode_lanes.argtypes = [vp, i32]
    lib.fw_model_decode_stats.argtypes = [vp, i64p, i64p, i64p, i32p]
    lib.fw_pack_blob_size.argtypes = [C.POINTER(FwConfig), C.POINTER(FwWeight), i32, i32, i64p, C.POINTER(vp)]
    lib.fw_pack_blob_copy.argtypes = [vp, vp, i64]
    lib.fw_pack_blob_free.argtypes = [vp]
    lib.fw_pack_blob_free.restype = None
    lib.fw_logmel.argtypes = [vp, vp, i64p, i32, vp, i32p]
    lib.fw_logmel_full.argtypes = [vp, vp, i64, vp, i64]
    lib.fw_encode.argtypes = [vp, vp, i32, C.POINTER(vp)]
    lib.fw_encode_pcm.argtypes = [vp, vp, i64p, i32, C.POINTER(vp)]
    lib.fw_encode_pcm_dev.argtypes = [vp, vp, i64p, i32, C.POINTER(vp)]
    lib.fw_tensor_shape.argtypes = [vp, i32p, i32p, i32p]
    lib.fw_tensor_to_host.argtypes = [vp, vp, vp]
    lib.fw_tensor_from_host.argtypes = [vp, vp, i32, C.POINTER(vp)]
    lib.fw_tensor_free.argtypes = [vp]
    lib.fw_tensor_free.restype = None
    lib.fw_generate.argtypes = [vp, vp, i32p, i32p, i32, C.POINTER(FwGenOpts), i32p, i32p, f32p, f32p]
    lib.fw_detect_language.argtypes = [vp, vp, i32, i32p, f32p]
    lib.fw_align.argtypes = [vp, vp, i32p, i32, i32p, i32p, i32p, i32, i32, i32, i32p, i32p, f32p]
    lib.fw_prof_enable.argtypes = [vp, i32]
    lib.fw_prof_enable.restype = None
    lib.fw_prof_reset.argtypes = [vp]
    lib.fw_prof_reset.restype = None
    lib.fw_prof_count.restype = i32
    lib.fw_prof_name.argtypes = [i32]
    lib.fw_prof_name.restype = C.c_char_p
    lib.fw_prof_get.argtypes = [vp, i32, C.POINTER(C.c_double), i64p, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    lib.fw_synchronize.argtypes = [vp]
    lib.fw_dev_alloc.argtypes = [vp, i64, C.POINTER(vp)]
    lib.fw_dev_free.argtypes = [vp, vp]
    lib.fw_dev_upload.argtypes = [vp, vp, vp, i64]
    lib.fw_test_gemm.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]
    lib.fw_test_dec_linear.argtypes = [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp, vp]
    lib.fw_test_dec_logits.argtypes = [vp, vp, i32, vp]
    lib.fw_test_logits_rules.argtypes = [vp, vp, i32, vp, i32, vp, C.POINTER(FwGenOpts), i32, vp, vp]
    lib.fw_bench_attention.argtypes = [vp, i32, i32, i32, i32, i32, f32p]
    lib.fw_bench_gemm.argtypes = [vp, i32, i32, i32, i32, i32, i32, i32, i32, f32p]
    lib.fw_bench_dec_linear.argtypes = [vp, i32, i32, i32, i32, i32, i32, f32p]
    lib.fw_test_layernorm.argtypes = [vp, vp, vp, vp, i32, i32, vp]
    lib.fw_test_attention.argtypes = [vp, vp, vp, vp, i32, i32, i32, vp]
    lib.fw_vad_create.argtypes = [C.POINTER(FwVadWeights), C.POINTER(vp)]
    lib.fw_vad_forward.argtypes = [vp, vp, i64, i32, vp, vp, vp]
    lib.fw_vad_free.argtypes = [vp]
    lib.fw_vad_forward_dev.argtypes = [vp, i32, vp, i64, vp, vp, vp]
    if hasattr(lib, "fw_vad_forward_audio_dev"):
        lib.fw_vad_forward_audio_dev.argtypes = [vp, i32, vp, i64, vp, vp, vp]
    lib.fw_vad_free.restype = None
    _lib = lib
    return lib


def check(rc: int):
    """Map a status code to the exception types CTranslate2 raises (ValueError / RuntimeError)."""
    if rc == FW_OK:
        return
    msg = load().fw_last_error().decode("utf-8", "replace")
    if rc in (FW_EINVAL, FW_ENOSPC):      # (a too-small caller buffer is a bad argument of that call)
        raise ValueError(msg)
    raise RuntimeError(f"libfwamd error {rc}: {msg}")


def ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def as_i32p(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def as_i64p(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_int64))


def as_f32p(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_float))
