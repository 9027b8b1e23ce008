"""ctypes binding of `include/mbistft_vits.h` (the C-ABI of the HIP path).

There is deliberately no fallback: if the shared library is missing or a call
fails, an exception is raised.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
# MBV_LIB: another build of the same library (A/B measurements of compile-time switches, scripts/stage_ab.py)
LIB_PATH = os.environ.get("MBV_LIB") or os.path.join(CSRC, "libmbistft_vits.so")

# every symbol include/mbistft_vits.h declares
SYMBOLS = [
    "mbv_abi_version", "mbv_create", "mbv_destroy", "mbv_last_error", "mbv_load_weight",
    "mbv_finalize_weights", "mbv_missing_weights", "mbv_encode", "mbv_synthesize", "mbv_decode",
    "mbv_speaker_embedding", "mbv_stage_times_ms", "mbv_istft_pqmf", "mbv_read_stage",
    "mbv_op_conv1d", "mbv_kernel_times_ms", "mbv_istft_finalize", "mbv_pcm16", "mbv_voice_conversion",
    "mbv_set_option", "mbv_arena_floats", "mbv_export_arena", "mbv_import_arena", "mbv_ticket", "mbv_stage_times_ms_at", "mbv_op_rel_attention",
]


ABI_VERSION = 3             # MBV_ABI_VERSION of include/mbistft_vits.h


class MbvConfig(C.Structure):
    _fields_ = [
        ("struct_bytes", C.c_int32), ("n_vocab", C.c_int32), ("inter_channels", C.c_int32),
        ("hidden_channels", C.c_int32), ("filter_channels", C.c_int32), ("n_heads", C.c_int32),
        ("n_layers", C.c_int32), ("kernel_size", C.c_int32),
        ("upsample_initial_channel", C.c_int32), ("spec_channels", C.c_int32),
        ("resblock_kernel_sizes", C.c_int32 * 3), ("resblock_dilations", (C.c_int32 * 3) * 3),
        ("resblock_type", C.c_int32),
        ("n_speakers", C.c_int32), ("gin_channels", C.c_int32), ("decoder", C.c_int32),
        ("device", C.c_int32), ("use_sdp", C.c_int32),
    ]


class MbvOutputs(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in
                ("o", "o_mb", "spec", "phase", "attn", "y_mask", "z", "z_p", "m_p", "logs_p")]


def build(force=False, verbose=False):
    """Compile the HIP sources for gfx950 in-tree (`csrc/build.sh`)."""
    if force:
        for f in os.listdir(CSRC):
            if f.endswith(".o") or f.endswith(".so"):
                os.remove(os.path.join(CSRC, f))
    r = subprocess.run(["bash", os.path.join(CSRC, "build.sh")], capture_output=True, text=True)
    if verbose or r.returncode:
        print(r.stdout, r.stderr)
    if r.returncode:
        raise RuntimeError("building libmbistft_vits.so failed:\n" + r.stderr[-4000:])
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise RuntimeError("%s not found: build it with `python -c 'import __graft_entry__ as g; "
                           "g.build()'` (hipcc, gfx950). There is no CPU fallback." % LIB_PATH)
    # torch first: it ships its own libamdhip64; loading this library before torch would bring in the
    # system HIP runtime as a second instance, which then sees no device ("no HIP device visible")
    import torch  # noqa: F401
    L = C.CDLL(LIB_PATH)
    vp, i32, i64p, fp = C.c_void_p, C.c_int, C.POINTER(C.c_int64), C.c_void_p
    L.mbv_abi_version.restype = i32
    L.mbv_create.argtypes = [C.POINTER(MbvConfig), C.POINTER(vp)]
    L.mbv_destroy.argtypes = [vp]
    L.mbv_destroy.restype = None
    L.mbv_last_error.argtypes = [vp]
    L.mbv_last_error.restype = C.c_char_p
    L.mbv_load_weight.argtypes = [vp, C.c_char_p, vp, i64p, i32]
    L.mbv_finalize_weights.argtypes = [vp, vp]
    L.mbv_missing_weights.argtypes = [vp, C.c_char_p, C.c_size_t]
    L.mbv_encode.argtypes = [vp, vp, vp, vp, i32, i32, C.c_float, vp, C.c_float, vp, vp]
    L.mbv_synthesize.argtypes = [vp, i32, vp, C.c_float, i32, C.POINTER(MbvOutputs), vp]
    L.mbv_decode.argtypes = [vp, vp, vp, i32, i32, C.POINTER(MbvOutputs), vp]
    L.mbv_speaker_embedding.argtypes = [vp, vp, i32, vp, vp]
    L.mbv_stage_times_ms.argtypes = [vp, C.POINTER(C.c_float * 5)]
    L.mbv_kernel_times_ms.argtypes = [vp, C.POINTER(C.c_float * 2)]
    L.mbv_istft_pqmf.argtypes = [vp, vp, i32, i32, vp, i32, vp, vp, vp, vp, vp]
    L.mbv_istft_finalize.argtypes = [vp, vp, vp, i32, i32, vp, vp, vp]
    L.mbv_voice_conversion.argtypes = [vp, vp, vp, vp, vp, i32, i32, vp, C.POINTER(MbvOutputs), vp, vp]
    L.mbv_pcm16.argtypes = [vp, vp, vp, i32, C.c_int64, i32, vp, vp]
    L.mbv_set_option.argtypes = [vp, C.c_char_p, i32]
    L.mbv_ticket.argtypes = [vp]
    L.mbv_ticket.restype = C.c_int64
    L.mbv_stage_times_ms_at.argtypes = [vp, C.c_int64, C.POINTER(C.c_float * 5)]
    L.mbv_arena_floats.argtypes = [vp]
    L.mbv_arena_floats.restype = C.c_int64
    L.mbv_export_arena.argtypes = [vp, vp, C.c_int64, vp]
    L.mbv_import_arena.argtypes = [vp, vp, C.c_int64, vp]
    L.mbv_read_stage.argtypes = [vp, C.c_char_p, vp, C.c_int64, vp]
    L.mbv_read_stage.restype = C.c_int64
    L.mbv_op_rel_attention.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp]
    L.mbv_op_conv1d.argtypes = [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, C.c_float, vp]
    for s in SYMBOLS:
        getattr(L, s)          # AttributeError if the header and the library ever drift
    if L.mbv_abi_version() != ABI_VERSION:
        raise RuntimeError("libmbistft_vits.so ABI version mismatch")
    _lib = L
    return L


class MbvError(RuntimeError):
    pass


def check(handle, rc, what):
    if rc:
        msg = lib().mbv_last_error(handle)
        raise MbvError("%s failed: %s" % (what, msg.decode() if msg else "unknown error"))
