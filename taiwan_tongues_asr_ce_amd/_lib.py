"""ctypes binding of libttasr.so (include/ttasr.h).  Loads the in-tree build only; there is no fallback:
a missing library or a missing GPU raises."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libttasr.so")

# every symbol include/ttasr.h declares (tests check the .so exports exactly these)
SYMBOLS = [
    "ttasr_create", "ttasr_create_shared", "ttasr_destroy", "ttasr_last_error", "ttasr_version", "ttasr_load_tensor", "ttasr_load_tensor_device",
    "ttasr_finalize_weights", "ttasr_log_mel", "ttasr_log_mel_windows", "ttasr_set_mel", "ttasr_encode", "ttasr_set_encoder_output",
    "ttasr_get_cross_kv", "ttasr_set_audio_ctx", "ttasr_generate", "ttasr_generate_capped", "ttasr_generate_beam", "ttasr_generate_beam_ragged", "ttasr_generate_sample", "ttasr_decode_reset", "ttasr_decode_step", "ttasr_apply_rules", "ttasr_align", "ttasr_dtw",
    "ttasr_set_option", "ttasr_phase_ms", "ttasr_beam_profile", "ttasr_encoder_kernel_ms", "ttasr_bench_kernel", "ttasr_bench_kernel_signature", "ttasr_sync",
]


class Config(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "n_mels", "n_audio_ctx", "d_model", "n_heads", "ffn_dim", "enc_layers", "dec_layers", "vocab",
        "n_text_ctx", "compute_type", "max_batch", "reserved")]


class GenOpts(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "max_new_tokens", "eot", "no_timestamps", "timestamp_begin", "no_speech", "sot_index", "timestamps",
        "max_initial_timestamp_index", "suppress_eot", "n_suppress", "n_begin_suppress", "check_interval")] + [
        ("suppress", C.POINTER(C.c_int32)), ("begin_suppress", C.POINTER(C.c_int32))]


def build(verbose: bool = False) -> str:
    """Compile libttasr.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    cmd = ["make", "-C", os.path.join(HERE, "csrc"), "-j", str(min(8, os.cpu_count() or 1))]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("libttasr build failed:\n" + r.stdout[-4000:] + r.stderr[-4000:])
    if verbose:
        print(r.stdout[-2000:])
    return LIB_PATH


_lib: Optional[C.CDLL] = None


def load() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(the HIP extension is the only implementation; there is no CPU fallback)")
    lib = C.CDLL(LIB_PATH)
    vp, i32, i64, f32p = C.c_void_p, C.c_int32, C.c_int64, C.POINTER(C.c_float)
    i32p, i64p = C.POINTER(C.c_int32), C.POINTER(C.c_int64)
    lib.ttasr_create.argtypes = [C.POINTER(Config), C.c_int, C.POINTER(vp)]
    lib.ttasr_create_shared.argtypes = [vp, i32, C.POINTER(vp)]
    lib.ttasr_destroy.argtypes = [vp]
    lib.ttasr_destroy.restype = None
    lib.ttasr_last_error.argtypes = [vp]
    lib.ttasr_last_error.restype = C.c_char_p
    lib.ttasr_version.argtypes = []
    lib.ttasr_version.restype = C.c_char_p
    lib.ttasr_load_tensor.argtypes = [vp, C.c_char_p, vp, i64p, i32]
    lib.ttasr_load_tensor_device.argtypes = [vp, C.c_char_p, vp, i32, i64p, i32]
    lib.ttasr_finalize_weights.argtypes = [vp]
    lib.ttasr_log_mel.argtypes = [vp, vp, i64, i64p, i32, i32, vp]
    lib.ttasr_log_mel_windows.argtypes = [vp, C.POINTER(vp), i64p, i64p, i32, vp, vp, vp]
    lib.ttasr_set_mel.argtypes = [vp, vp, i32]
    lib.ttasr_encode.argtypes = [vp, i32, vp]
    lib.ttasr_set_encoder_output.argtypes = [vp, vp, i32]
    lib.ttasr_get_cross_kv.argtypes = [vp, i32, i32, i32, vp]
    lib.ttasr_set_audio_ctx.argtypes = [vp, i32]
    lib.ttasr_generate.argtypes = [vp, i32, i32p, i32p, i32, C.POINTER(GenOpts), i32p, i32p, f32p, f32p]
    lib.ttasr_generate_capped.argtypes = [vp, i32, i32p, i32p, i32, C.POINTER(GenOpts), i32p, i32p, i32p, f32p, f32p]
    lib.ttasr_generate_beam.argtypes = [vp, i32, i32, i32p, i32, C.POINTER(GenOpts), C.c_float, i32p, i32p, f32p, f32p]
    lib.ttasr_generate_beam_ragged.argtypes = [vp, i32, i32, i32p, i32p, i32p, i32, C.POINTER(GenOpts), C.c_float, i32p, i32p, f32p, f32p]
    lib.ttasr_generate_sample.argtypes = [vp, i32, i32, i32p, i32, C.POINTER(GenOpts), C.c_float, C.c_uint32, i32p, i32p, f32p, f32p]
    lib.ttasr_decode_reset.argtypes = [vp, i32]
    lib.ttasr_decode_step.argtypes = [vp, i32p, i32, vp]
    lib.ttasr_apply_rules.argtypes = [vp, vp, i32p, i32, i32, C.POINTER(GenOpts), vp, i32p]
    lib.ttasr_align.argtypes = [vp, i32, i32p, i32, i32p, i32, f32p, f32p]
    lib.ttasr_dtw.argtypes = [f32p, i32, i32, i32p, i32p, i32p]
    lib.ttasr_set_option.argtypes = [vp, C.c_char_p, i32]
    lib.ttasr_phase_ms.argtypes = [vp, f32p]
    lib.ttasr_beam_profile.argtypes = [vp, f32p]
    lib.ttasr_encoder_kernel_ms.argtypes = [vp, f32p]
    lib.ttasr_bench_kernel.argtypes = [vp, C.c_char_p, i32, i32, f32p, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    lib.ttasr_sync.argtypes = [vp]
    for s in SYMBOLS:
        f = getattr(lib, s)
        if s not in ("ttasr_destroy", "ttasr_last_error", "ttasr_version"):
            f.restype = C.c_int
    _lib = lib
    return lib
