"""ctypes binding of csrc/libdj_hip.so (the C ABI declared in include/dj_hip.h).

There is no CPU fallback: if the library is missing the import of any compute entry point
raises, and every call checks the return code and raises `DjError` with dj_last_error()."""
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_double, c_float, c_int, c_long, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DJ_LIB_PATH") or os.path.join(_HERE, "csrc", "libdj_hip.so")   # override: kernel experiments


class DjError(RuntimeError):
    pass


class BnTrain(Structure):   # dj_bn_train
    _fields_ = [("acc", c_void_p), ("ticket", c_void_p), ("gamma", c_void_p), ("beta", c_void_p),
                ("moving_mean", c_void_p), ("moving_var", c_void_p), ("scale", c_void_p), ("shift", c_void_p),
                ("save_mean", c_void_p), ("save_invstd", c_void_p), ("eps", c_float), ("momentum", c_float)]


class CopyPart(Structure):   # dj_copy_part
    _fields_ = [("src", c_void_p), ("dst", c_void_p), ("ld_src", c_long), ("ld_dst", c_long), ("rows", c_long),
                ("cols", c_long), ("beta", c_int)]


class ColsumPart(Structure):   # dj_colsum_part
    _fields_ = [("x", c_void_p), ("out", c_void_p), ("rows", c_long), ("C", c_int), ("ld", c_int), ("beta", c_int)]


class ConvDesc(Structure):
    """Mirror of `dj_conv2d_desc` (include/dj_hip.h)."""
    _fields_ = [(n, c_int) for n in (
        "batch", "in_h", "in_w", "in_c", "out_h", "out_w", "out_c",
        "kernel_h", "kernel_w", "stride_h", "stride_w", "dilation_h", "dilation_w",
        "pad_top", "pad_left", "ld_x", "ld_y")]


_lib = None
ABI_VERSION = 3     # DJ_ABI_VERSION of include/dj_hip.h these bindings were written for


def load():
    """Load the shared library (idempotent).  Raises DjError when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DjError(
            "HIP extension %s is missing -- run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(there is no CPU fallback for the compute path)" % LIB_PATH)
    # torch first: its wheel bundles the HIP runtime (libamdhip64) the process must share -- loading this library
    # before torch would bind it to /opt/rocm's copy instead, and kernels registered with one runtime cannot be
    # launched on the device the other one owns ("no ROCm-capable device is detected")
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    lib.dj_abi_version.restype = c_int
    have = lib.dj_abi_version()
    if have != ABI_VERSION:
        # a stale build product (the .so is git-ignored and travels with the working tree): calling it through these
        # signatures would pass arguments it reads differently
        raise DjError("%s reports ABI version %d, these bindings are written for %d: rebuild it "
                      "(`python -c 'import __graft_entry__ as g; g.build()'`)" % (LIB_PATH, have, ABI_VERSION))
    _declare(lib)
    _lib = lib
    return lib


# name -> (restype, [argtypes]); kept in one table so the "exports every symbol" test can walk it
FP = c_void_p  # device float* passed as integer address
SIGNATURES = {
    "dj_last_error": (c_char_p, []),
    "dj_abi_version": (c_int, []),
    "dj_conv2d_fwd_stats_rows": (c_int, [POINTER(ConvDesc)]),
    "dj_conv2d_nhwc_fwd": (c_int, [POINTER(ConvDesc), FP, FP, FP, FP, FP, FP, c_int, c_int, FP, c_void_p]),
    "dj_conv2d_nhwc_fwd_ws": (c_int, [POINTER(ConvDesc), FP, FP, FP, FP, FP, FP, c_int, c_int, FP, FP, c_long, c_void_p]),
    "dj_conv2d_fwd_workspace_floats": (c_long, [POINTER(ConvDesc), c_int]),
    "dj_conv2d_nhwc_dgrad": (c_int, [POINTER(ConvDesc), FP, FP, FP, FP, c_int, c_void_p]),
    "dj_conv2d_nhwc_dgrad_bnbwd": (c_int, [POINTER(ConvDesc), FP, FP, FP, FP, c_int, FP, FP, FP, FP, FP, c_void_p]),
    "dj_conv2d_nhwc_wgrad": (c_int, [POINTER(ConvDesc), FP, FP, FP, FP, FP, c_int, c_int, c_void_p]),
    "dj_conv2d_nhwc_fwd_t": (c_int, [POINTER(ConvDesc), FP, c_int, FP, c_int, FP, FP, c_int, FP, FP, c_int, c_int, FP, FP, c_int, FP,
                                     FP, FP, c_int, c_int, FP, c_long, c_void_p]),
    "dj_conv2d_nhwc_dgrad_t": (c_int, [POINTER(ConvDesc), FP, c_int, FP, c_int, FP, FP, c_int, c_int, FP, c_int, c_int, FP, FP, FP,
                                       FP, FP, c_void_p]),
    "dj_shadow_weights": (c_int, [FP, FP, FP, c_long, c_void_p]),
    "dj_conv2d_nhwc_wgrad_t": (c_int, [POINTER(ConvDesc), FP, c_int, FP, c_int, FP, FP, FP, c_int, c_int, c_void_p]),
    "dj_affine_act_t": (c_int, [FP, c_int, c_int, FP, FP, FP, c_int, c_int, FP, FP, FP, c_int, c_int, c_long, c_int, c_int,
                                c_void_p]),
    "dj_bn_bwd_reduce_t": (c_int, [FP, c_int, c_int, FP, c_int, c_int, FP, c_int, c_int, FP, FP, FP, FP, c_int, c_long, c_int, FP,
                                   c_void_p]),
    "dj_bn_bwd_apply_t": (c_int, [FP, c_int, c_int, FP, c_int, c_int, FP, c_int, c_int, FP, FP, c_int, FP, FP, FP, FP, c_int,
                                  c_int, c_long, c_int, FP, c_int, c_int, c_int, c_void_p]),
    "dj_relu_bwd_t": (c_int, [FP, c_int, c_int, FP, c_int, c_int, FP, c_int, c_int, c_long, c_int, c_int, c_void_p]),
    "dj_copy2d_t": (c_int, [FP, c_int, c_long, FP, c_int, c_long, c_long, c_long, c_int, c_void_p]),
    "dj_conv2d_fwd_addrelu_supported": (c_int, [POINTER(ConvDesc)]),
    "dj_conv2d_nhwc_fwd_addrelu": (c_int, [POINTER(ConvDesc), FP, FP, FP, FP, FP, FP, FP, c_int, FP, FP, FP, c_int, c_int, FP,
                                           c_void_p]),
    "dj_conv2d_nhwc_fwd_addrelu_ws": (c_int, [POINTER(ConvDesc), FP, FP, FP, FP, FP, FP, FP, c_int, FP, FP, FP, c_int, c_int,
                                              FP, FP, c_long, c_void_p]),
    "dj_colsum_direct": (c_int, [FP, c_long, c_int, c_int, FP, c_int, c_void_p]),
    "dj_colsum_multi": (c_int, [POINTER(ColsumPart), c_int, c_void_p]),
    "dj_copy2d_multi": (c_int, [POINTER(CopyPart), c_int, c_void_p]),
    "dj_conv2d_nhwc_fwd_bn": (c_int, [POINTER(ConvDesc), FP, FP, FP, FP, FP, FP, c_int, FP, c_int, FP, FP, FP, c_int,
                                      POINTER(BnTrain), c_void_p]),
    "dj_set_fast_path": (None, [c_int]),
    "dj_set_compute_mode": (c_int, [c_int]),
    "dj_set_thread_compute_mode": (c_int, [c_int]),
    "dj_get_compute_mode": (c_int, []),
    "dj_conv2d_tune_configs": (c_int, []),
    "dj_conv2d_tune_set": (c_int, [c_int, POINTER(ConvDesc), c_int, c_int]),
    "dj_conv2d_default_config": (c_int, [c_int, POINTER(ConvDesc), POINTER(c_int), POINTER(c_int)]),
    "dj_reduce_rows": (c_int, [c_long]),
    "dj_colstats_partial": (c_int, [FP, c_long, c_int, c_int, FP, c_void_p]),
    "dj_colsum_partial": (c_int, [FP, c_long, c_int, c_int, FP, c_void_p]),
    "dj_colreduce_finalize": (c_int, [FP, c_int, c_int, c_int, FP, c_int, c_void_p]),
    "dj_bn_train_finalize": (c_int, [FP, c_int, c_long, FP, FP, FP, c_float, c_float, FP, FP, FP, FP, FP, FP, c_int,
                                     c_void_p]),
    "dj_bn_infer_coeffs": (c_int, [FP, FP, FP, FP, c_float, FP, FP, c_int, c_void_p]),
    "dj_affine_act": (c_int, [FP, c_int, FP, FP, FP, c_int, FP, FP, FP, c_int, c_long, c_int, c_int, c_void_p]),
    "dj_bn_bwd_reduce": (c_int, [FP, c_int, FP, c_int, FP, c_int, FP, FP, FP, FP, c_int, c_long, c_int, FP, c_void_p]),
    "dj_bn_bwd_finalize": (c_int, [FP, c_int, c_long, FP, FP, FP, FP, FP, FP, FP, FP, c_int, c_void_p]),
    "dj_bn_bwd_apply": (c_int, [FP, c_int, FP, c_int, FP, c_int, FP, FP, c_int, FP, FP, FP, FP, c_int, c_long, c_int,
                                FP, c_int, c_int, c_void_p]),
    "dj_relu_bwd": (c_int, [FP, c_int, FP, c_int, FP, c_int, c_long, c_int, c_int, c_void_p]),
    "dj_copy2d": (c_int, [FP, c_long, FP, c_long, c_long, c_long, c_int, c_void_p]),
    "dj_upsample2x": (c_int, [FP, c_int, FP, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "dj_l2norm_fwd": (c_int, [FP, c_int, FP, FP, c_int, FP, c_long, c_int, c_void_p]),
    "dj_l2norm_bwd": (c_int, [FP, c_int, FP, c_int, FP, FP, FP, c_int, FP, c_long, c_int, c_int, c_void_p]),
    "dj_maxpool2d_fwd": (c_int, [FP, FP] + [c_int] * 13 + [c_void_p, c_void_p]),
    "dj_maxpool2d_bwd": (c_int, [FP, FP, FP] + [c_int] * 14 + [c_void_p, c_void_p]),
    "dj_softmax_fwd": (c_int, [FP, FP, c_long, c_int, c_void_p]),
    "dj_softmax_bwd": (c_int, [FP, FP, c_long, FP, c_long, c_int, c_int, c_void_p]),
    "dj_ssd_loss_workspace_floats": (c_long, [c_long]),
    "dj_ssd_loss_fwd": (c_int, [FP, FP, c_long, c_int, c_int, c_int, c_float, FP, FP, c_void_p]),
    "dj_ssd_loss_bwd": (c_int, [FP, FP, c_long, c_int, c_float, c_float, FP, FP, FP, c_void_p]),
    "dj_categorical_crossentropy": (c_int, [FP, FP, c_long, c_int, c_float, FP, FP, FP, c_void_p]),
    "dj_sgd_momentum_update": (c_int, [FP, FP, FP, c_long, c_float, c_float, c_int, c_float, c_float, FP, c_void_p]),
    "dj_decode_detections_workspace_floats": (c_long, [c_int, c_int, c_int, c_int]),
    "dj_decode_detections": (c_int, [FP, c_int, c_int, c_int, c_float, c_float, c_int, c_int, c_int, c_int, c_int, FP, FP,
                                     c_void_p]),
    "dj_decode_detections_fast_workspace_floats": (c_long, [c_int, c_int, c_int]),
    "dj_decode_detections_fast": (c_int, [FP, c_int, c_int, c_int, c_float, c_float, c_int, c_int, c_int, c_int, c_int, FP, FP,
                                          c_void_p]),
    "dj_ssd_encode_targets": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                                      c_int, c_double, c_double, c_int, FP, c_void_p]),
    "dj_global_avg_pool_fwd": (c_int, [FP, FP, c_int, c_int, c_int, c_void_p]),
    "dj_global_avg_pool_bwd": (c_int, [FP, FP, c_int, c_int, c_int, c_int, c_void_p]),
}


def _declare(lib):
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args


def check(rc, what=""):
    if rc < 0:
        msg = load().dj_last_error()
        raise DjError("%s failed (rc=%d): %s" % (what, rc, msg.decode() if msg else "?"))
    return rc


def ptr(t):
    """Device address of a torch tensor (None -> NULL)."""
    if t is None:
        return None
    return t.data_ptr()


# The HIP stream the next launch goes to.  engine.Plan sets it while it runs its launch lists (the main stream, and the side
# stream around the launches it issues there): a plain attribute instead of torch's stream context manager, whose enter /
# exit cost ~25 us of host time per side-stream launch -- in the stretches of the step that consist of 5-10 us kernels (the
# extra-feature layers and predictor heads) the host was what the GPU waited for.  None: torch's current stream.
import threading

_stream_tls = threading.local()      # per thread: two threads may run plans of their own side by side


def set_launch_stream(handle):
    """Set (None: clear) the calling thread's launch stream; -> the previous setting."""
    prev = getattr(_stream_tls, "handle", None)
    _stream_tls.handle = handle
    return prev


def current_stream():
    h = getattr(_stream_tls, "handle", None)
    if h is not None:
        return h
    import torch
    return torch.cuda.current_stream().cuda_stream
