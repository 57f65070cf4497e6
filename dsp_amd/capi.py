"""ctypes binding of libwpdsp.so (include/wpdsp.h). One Python function per C
entry point; errors become exceptions of the class the reference would throw:

* WP_ERR_RANGE     -> IndexError   (RangeError from TypedArray.set, lib/convolve.js:590)
* WP_ERR_FFT_SIZE  -> ValueError   (Error('FFT size must be ...'), lib/convolve.js:8-9)
* everything else  -> RuntimeError
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libwpdsp.so")

WP_ERR_INVALID, WP_ERR_RANGE, WP_ERR_FFT_SIZE, WP_ERR_NO_DEVICE, WP_ERR_HIP, WP_ERR_UNSUPPORTED = -1, -2, -3, -4, -5, -6

f32p = C.POINTER(C.c_float)
f64p = C.POINTER(C.c_double)
i32p = C.POINTER(C.c_int32)
i64p = C.POINTER(C.c_int64)


class WpSrc(C.Structure):
    """struct wp_src (include/wpdsp.h)"""
    _fields_ = [("ch", C.c_void_p * 2), ("nch", C.c_int32), ("flags", C.c_int32), ("rl", C.c_int64),
                ("sl", C.c_int64), ("vol", C.c_double), ("vol_vec", C.c_void_p), ("o", C.c_int64)]


class WpReverbView(C.Structure):
    """struct wp_reverb_view (include/wpdsp.h)"""
    _fields_ = [("curr_len", C.c_int64), ("ret_off", C.c_int64), ("ret_len", C.c_int64),
                ("prev_off", C.c_int64), ("prev_len", C.c_int64)]


# name -> (restype, argtypes); this table is also what tests check against the header
PROTOTYPES = {
    "wp_init": (C.c_int, []),
    "wp_last_error": (C.c_char_p, []),
    "wp_next_pot": (C.c_int32, [C.c_int32]),
    "wp_fft_length": (C.c_int32, [C.c_int64, C.c_int64]),
    "wp_ir_create": (C.c_void_p, [C.POINTER(f32p), C.c_int, C.c_int64, C.c_int64]),
    "wp_ir_create_dev": (C.c_void_p, [C.c_void_p, C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_void_p]),
    "wp_ir_destroy": (None, [C.c_void_p]),
    "wp_ir_info": (C.c_int64, [C.c_void_p, C.c_int]),
    "wp_convolve": (C.c_int, [C.c_void_p, C.c_int, f32p, C.c_int64, f32p]),
    "wp_convolve_dev": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "wp_direct_convolution": (C.c_int, [f32p, C.c_int64, f32p, C.c_int64, f32p]),
    "wp_kernel_spectrum": (C.c_int, [f32p, C.c_int64, C.c_int64, f32p]),
    "wp_kernel_from_spectrum": (C.c_int, [f32p, C.c_int64, C.c_int64, f32p]),
    "wp_convolve_dev_noalloc": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "wp_conv_create": (C.c_void_p, [C.c_void_p, C.c_int, i32p]),
    "wp_conv_destroy": (None, [C.c_void_p]),
    "wp_conv_run_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64,
                                  C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]),
    "wp_conv_mix_run_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.POINTER(C.c_void_p), C.c_int64,
                                      f64p, C.c_void_p]),
    "wp_conv_run_mix_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64,
                                      C.POINTER(C.c_void_p), C.c_int64, f64p, C.c_void_p]),
    "wp_conv_workspace_bytes": (C.c_int64, [C.c_void_p]),
    "wp_conv_profile": (C.c_int, [C.c_void_p, C.c_int]),
    "wp_conv_profile_read": (C.c_int, [C.c_void_p, C.POINTER(C.c_float)]),
    "wp_conv_profile_sum": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int)]),
    "wp_conv_info": (C.c_int64, [C.c_void_p, C.c_int]),
    "wp_reverb_view_for": (C.c_int, [C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.POINTER(WpReverbView)]),
    "wp_reverb_create": (C.c_void_p, [C.c_void_p, C.c_int, i32p, C.c_int64]),
    "wp_reverb_destroy": (None, [C.c_void_p]),
    "wp_reverb_block": (C.c_int, [C.c_void_p, C.POINTER(f32p), C.c_int64, C.POINTER(f32p), C.POINTER(WpReverbView)]),
    "wp_reverb_block_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.POINTER(C.c_void_p), i64p,
                                      C.POINTER(WpReverbView), C.c_void_p]),
    "wp_reverb_block_to": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64,
                                     C.POINTER(WpReverbView), C.c_void_p]),
    "wp_reverb_set_tail": (C.c_int, [C.c_void_p, C.POINTER(f32p), C.c_int64]),
    "wp_reverb_set_tail_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64]),
    "wp_reverb_tail_len": (C.c_int64, [C.c_void_p]),
    "wp_reverb_get_tail": (C.c_int, [C.c_void_p, C.POINTER(f32p)]),
    "wp_mix_sl": (C.c_int32, [C.c_int64, C.c_double]),
    "wp_mix": (C.c_int, [C.POINTER(f32p), C.c_int, C.c_int64, C.POINTER(WpSrc), C.c_int64]),
    "wp_mix_dev": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int64, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]),
    "wp_comm_unique_id": (C.c_int, [C.c_void_p]),
    "wp_comm_create": (C.c_void_p, [C.c_void_p, C.c_int, C.c_int]),
    "wp_comm_destroy": (None, [C.c_void_p]),
    "wp_comm_info": (C.c_int64, [C.c_void_p, C.c_int]),
    "wp_comm_library": (C.c_char_p, []),
    "wp_comm_available": (C.c_int, []),
    "wp_host_alloc": (C.c_void_p, [C.c_size_t]),
    "wp_host_free": (None, [C.c_void_p]),
    "wp_host_pin": (C.c_int, [C.c_void_p, C.c_size_t]),
    "wp_host_unpin": (C.c_int, [C.c_void_p]),
    "wp_lf_cover": (C.c_int, [C.c_int64, C.c_int64, C.c_int, C.POINTER(C.c_int64)]),
    "wp_lf_wide_frames": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "wp_bus_reduce": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p]),
    "wp_bus_allreduce": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "wp_buf_create": (C.c_void_p, [C.c_int, C.c_int64]),
    "wp_buf_id": (C.c_uint64, [C.c_void_p]),
    "wp_buf_open": (C.c_void_p, [C.c_uint64]),
    "wp_buf_release": (None, [C.c_void_p]),
    "wp_buf_info": (C.c_int64, [C.c_void_p, C.c_int]),
    "wp_buf_channel": (C.c_void_p, [C.c_void_p, C.c_int]),
    "wp_buf_write": (C.c_int, [C.c_void_p, C.c_int, C.c_int64, f32p, C.c_int64]),
    "wp_buf_read": (C.c_int, [C.c_void_p, C.c_int, C.c_int64, f32p, C.c_int64]),
    "wp_buf_copy": (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_int, C.c_int64, C.c_int64, C.c_void_p]),
    "wp_buf_fill": (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_int64, C.c_float, C.c_void_p]),
    "wp_ring_pos": (C.c_int64, [C.c_int64, C.c_int64, C.c_int64]),
    "wp_ring_advance": (C.c_int64, [C.c_int64, C.c_int64]),
    "wp_beat_rate": (C.c_int64, [C.c_double, C.c_double]),
    "wp_bpm_from_beat_rate": (C.c_double, [C.c_double, C.c_int64]),
    "wp_tune": (C.c_int, [C.c_int, C.c_int64]),
    "wp_synth_noise_dev": (C.c_int, [C.c_void_p, C.c_int64, C.c_uint32, C.c_int64, C.c_void_p]),
    "wp_synth_ir": (C.c_int, [f32p, C.c_int64, C.c_uint32]),
}

_lib = None


def lib():
    """Load libwpdsp.so; raise loudly if it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("libwpdsp.so is not built: run `make -C dsp_amd/csrc` "
                               "(or __graft_entry__.build()); there is no CPU fallback")
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def last_error():
    return lib().wp_last_error().decode("utf-8", "replace")


def check(rc, what=""):
    """Turn a negative return code into the exception class the reference throws."""
    if rc is None:
        rc = WP_ERR_HIP
    if isinstance(rc, int) and rc >= 0:
        return rc
    msg = "%s: %s" % (what, last_error()) if what else last_error()
    if rc == WP_ERR_RANGE:
        raise IndexError(msg)
    if rc == WP_ERR_FFT_SIZE:
        raise ValueError(msg)
    if rc == WP_ERR_UNSUPPORTED:
        raise NotImplementedError(msg)
    raise RuntimeError("%s (code %d)" % (msg, rc))


def check_handle(h, what):
    if not h:
        msg = last_error()
        if "power of two" in msg or "nextPowerOfTwo" in msg:
            raise ValueError("%s: %s" % (what, msg))
        raise RuntimeError("%s: %s" % (what, msg))
    return h


def init():
    """Number of HIP devices; raises RuntimeError without a GPU."""
    return check(lib().wp_init(), "wp_init")
