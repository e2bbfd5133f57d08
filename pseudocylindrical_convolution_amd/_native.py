"""ctypes binding of the two C-ABI libraries (include/pconv_hip.h, pconv_coder.h).

There is no fallback: if libpconv_hip.so is missing or a call fails, PconvError is
raised.  Nothing here imports the oracle.
"""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_float, c_int, c_int32, c_longlong, c_size_t, c_uint8, c_uint32, c_void_p

HERE = os.path.dirname(os.path.abspath(__file__))


class PconvError(RuntimeError):
    pass


P = c_void_p  # device or host pointer passed as an integer address
I = c_int
F = c_float
LL = c_longlong

_HIP_SIGNATURES = {
    # host geometry
    "pconv_host_tile_widths": [P, I, I, I, P],
    "pconv_host_slice_taps": [P, I, I, P, P],
    "pconv_host_uslice_taps": [P, I, I, P, P],
    "pconv_host_pad_table": [P, I, I, I, I, P, P, P, P],
    "pconv_host_wavefront": [P, I, I, I, P, P],
    "pconv_host_causal_halo": [P, I, I, I, I, I, P, P, P, P, P, P],
    "pconv_host_project_table": [P, P, I, F, I, I, I, I, P],
    # transform path
    "pconv_sphere_slice": [P, P, P, P, P, I, I, I, I, I, I, P],
    "pconv_sphere_uslice": [P, P, P, P, P, I, I, I, I, I, I, P],
    "pconv_pseudo_pad": [P, P, P, P, P, P, P, I, I, I, I, I, I, P],
    "pconv_pseudo_pad_ring": [P, P, P, P, P, P, I, I, I, I, I, I, I, P],
    "pconv_pseudo_fill": [P, P, I, I, I, I, I, I, I, F, P],
    "pconv_dtow": [P, P, I, I, I, I, I, I, P],
    "pconv_quant": [P, P, P, P, P, P, P, I, I, I, I, I, I, P],
    "pconv_dquant": [P, P, P, P, P, I, I, I, I, I, I, I, P],
    "pconv_leaky_clip": [P, LL, P],
    "pconv_frames_u8_to_f32": [P, P, I, I, I, P],
    "pconv_frames_f32_to_u8": [P, P, I, I, I, P],
    "pconv_project": [P, P, P, I, I, I, I, I, I, I, I, P],
    "pconv_context_reshape": [P, P, I, I, I, I, I, P],
    "pconv_mask_constrain": [P, I, I, I, I, I, P],
    # backward of the linear geometry ops (training path)
    "pconv_context_reshape_backward": [P, P, I, I, I, I, I, P],
    "pconv_sphere_slice_backward": [P, P, P, P, P, I, I, I, I, I, I, P],
    "pconv_sphere_uslice_backward": [P, P, P, P, P, I, I, I, I, I, I, P],
    "pconv_host_pad_reverse": [P, I, I, I, I, P, P, P],
    "pconv_pseudo_pad_backward": [P, P, P, P, P, P, I, I, I, I, I, I, P],
    "pconv_quant_backward": [P, P, P, P, P, P, P, P, P, P, F, I, I, I, I, I, I, P],
    "pconv_project_backward": [P, P, P, P, I, I, I, I, I, I, I, I, P],
    "pconv_entropy_pad": [P, P, P, P, P, I, I, I, I, I, I, P],
    "pconv_host_entropy_pad_table": [P, I, I, I, I, I, P, P],
    "pconv_host_causal_reverse": [P, I, I, I, I, I, P, P, P],
    "pconv_entropy_pad_backward": [P, P, P, P, P, P, I, I, I, I, I, I, P],
    "pconv_gmm_loss": [P, P, P, P, P, P, P, P, P, I, I, P],
    "pconv_conv_packed_size": [I, I, I, P, P],
    "pconv_conv_pack_weight": [P, P, I, I, I, P],
    "pconv_conv2d": [P, P, P, P, I, I, I, I, I, I, I, I, P, P, I, P, P, I, I, P, P],
    "pconv_gdn": [P, P, P, P, I, I, I, I, I, P, I, P, P, P],
    "pconv_wino_packed_size": [I, I],
    "pconv_wino_pack_weight": [P, P, I, I, P],
    "pconv_wino_supported": [I, I, I, I, I],
    "pconv_conv3x3_wino": [P, P, P, P, I, I, I, I, I, I, P, P, I, P, I, I, P, P],
    "pconv_conv3x3_wino_flat": [P, P, P, P, I, I, I, I, I, I, P, P, I, P, I, I, P, P],
    "pconv_wino42_packed_size": [I, I],
    "pconv_wino42_pack_weight": [P, P, I, I, P],
    "pconv_wino42_supported": [I, I, I, I, I],
    "pconv_conv3x3_wino42": [P, P, P, P, I, I, I, I, I, I, P, P, I, P, I, I, P, P],
    # entropy wavefront
    "pconv_dinput2": [P, P, P, I, I, I, I, I, I, I, I, I, F, I, P],
    "pconv_ctx_pad_run2": [P, P, P, P, P, P, I, I, I, I, I, I, I, I, I, I, P],
    "pconv_entropy_conv": [P, P, P, P, P, P, P, I, I, I, I, I, I, I, I, I, I, I, I, I, I, I, I, P, P, P, P, P],
    "pconv_host_causal_table": [P, I, I, I, I, P, P],
    "pconv_entropy_add": [P, P, P, I, I, I, I, I, I, I, I, I, I, P],
    "pconv_dextract2": [P, P, P, I, I, I, I, I, I, I, I, I, P],
    "pconv_dextract2_batch": [P, P, P, I, I, I, I, I, I, I, I, I, I, LL, P],
    "pconv_gmm_table": [P, P, P, P, I, I, I, F, F, F, I, P],
    "pconv_ctx_to_symbols": [P, P, P, I, I, I, I, I, I, F, P],
    "pconv_symbols_to_ctx": [P, P, P, I, I, I, I, I, I, F, I, P],
    "pconv_step_tables": [P, P, P, P, P, I, I, I, I, I, I, I, I, I, F, F, F, P],
    # native entropy engine
    "pconv_ee_set_layer": [P, I, P, P, P, P],
    "pconv_ee_steps": [P],
    "pconv_ee_host_cpus": [],
    "pconv_ee_spin_us": [I],
    "pconv_ee_host_plan": [I, P, P, P, P],
    "pconv_ee_wait_mode": [P],
    "pconv_device_blocking_sync": [I],
    "pconv_stream_create": [P],
    "pconv_stream_destroy": [P],
    "pconv_ee_encode": [P, P, P],
    "pconv_ee_encode_begin": [P, P, P],
    "pconv_ee_set_encode_ranges": [P, I],
    "pconv_ee_encode_end": [P, P],
    "pconv_ee_decode": [P, P, P, P, P],
}

_CODER_SIGNATURES = {
    "pconv_coder_new": ([c_char_p], c_void_p),
    "pconv_coder_free": ([c_void_p], None),
    "pconv_coder_error": ([c_void_p], c_char_p),
    "pconv_coder_start_encoder": ([c_void_p], c_int),
    "pconv_coder_encode": ([c_void_p, P, c_uint32, c_uint32, c_uint32], c_int),
    "pconv_coder_encodes": ([c_void_p, P, c_int, P, c_int], c_int),
    "pconv_coder_end_encoder": ([c_void_p], c_int),
    "pconv_coder_bytes": ([c_void_p, POINTER(c_size_t)], POINTER(c_uint8)),
    "pconv_coder_start_decoder": ([c_void_p], c_int),
    "pconv_coder_start_decoder_mem": ([c_void_p, P, c_size_t], c_int),
    "pconv_coder_decode": ([c_void_p, P, c_uint32, c_uint32], c_int),
    "pconv_coder_decodes": ([c_void_p, P, c_int, P, c_int], c_int),
    "pconv_coder_decodes_i32": ([c_void_p, P, c_int, P, c_int], c_int),
    "pconv_coder_encodes_rows16": ([c_void_p, P, c_int], c_int),
    "pconv_coder_decodes_rows16_i32": ([c_void_p, P, P, c_int], c_int),
}

_hip = None
_coder = None


def _load(name):
    path = os.path.join(HERE, name)
    if name == "libpconv_hip.so" and os.environ.get("PCONV_HIP_LIB"):
        path = os.environ["PCONV_HIP_LIB"]  # kernel tuning experiments only
    if not os.path.exists(path):
        raise PconvError(
            "%s is not built: run `python -m pseudocylindrical_convolution_amd.build` "
            "(needs hipcc, targets gfx950); there is no CPU fallback" % path)
    return ctypes.CDLL(path)


def hip_lib():
    """libpconv_hip.so with typed entry points; raises PconvError when absent."""
    global _hip
    if _hip is None:
        lib = _load("libpconv_hip.so")
        for fn, args in _HIP_SIGNATURES.items():
            f = getattr(lib, fn)
            f.argtypes = args
            f.restype = c_int
        lib.pconv_last_error.restype = c_char_p
        lib.pconv_last_error.argtypes = []
        lib.pconv_abi_version.restype = c_int
        lib.pconv_device_count.restype = c_int
        lib.pconv_ee_create.argtypes = [I, I, I, I, I, P, F, I, F, F]
        lib.pconv_ee_create.restype = c_void_p
        lib.pconv_ee_destroy.argtypes = [P]
        lib.pconv_ee_destroy.restype = None
        lib.pconv_ee_symbols_per_image.argtypes = [P]
        lib.pconv_ee_symbols_per_image.restype = c_longlong
        lib.pconv_ee_stream.argtypes = [P, I, POINTER(c_size_t)]
        lib.pconv_ee_stream.restype = POINTER(c_uint8)
        lib.pconv_wino_packed_size.restype = c_longlong
        lib.pconv_wino42_packed_size.restype = c_longlong
        _hip = lib
    return _hip


def coder_lib():
    global _coder
    if _coder is None:
        lib = _load("libpconv_coder.so")
        for fn, (args, res) in _CODER_SIGNATURES.items():
            f = getattr(lib, fn)
            f.argtypes = args
            f.restype = res
        _coder = lib
    return _coder


def declared_hip_symbols():
    return sorted(list(_HIP_SIGNATURES) + ["pconv_last_error", "pconv_abi_version", "pconv_device_count",
                                           "pconv_ee_create", "pconv_ee_destroy", "pconv_ee_symbols_per_image",
                                           "pconv_ee_stream"])


def declared_coder_symbols():
    return sorted(_CODER_SIGNATURES)


def check(rc, what=""):
    """Raise PconvError for a negative status of a libpconv_hip call."""
    if rc < 0:
        msg = hip_lib().pconv_last_error()
        raise PconvError("%s failed (%d): %s" % (what or "pconv call", rc, (msg or b"").decode()))
    return rc


def call(fn, *args):
    lib = hip_lib()
    return check(getattr(lib, fn)(*args), fn)
