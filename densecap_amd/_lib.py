"""ctypes binding of libdensecap_hip.so (the C ABI declared in include/densecap.h; measurement / test hooks in
include/densecap_debug.h).

There is deliberately NO fallback: if the shared library has not been built
(`python -c "import __graft_entry__ as g; g.build()"` or `make -C densecap_amd/csrc`)
importing the product path raises, and if no HIP device is present dc_create fails.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# DENSECAP_HIP_LIB: another build of the same library (the LuaJIT binding honours the same variable)
LIB_PATH = os.environ.get("DENSECAP_HIP_LIB") or os.path.join(_HERE, "lib", "libdensecap_hip.so")

DC_NUM_VGG_CONVS = 13
c_float_p = C.POINTER(C.c_float)
c_int32_p = C.POINTER(C.c_int32)


class DcWeights(C.Structure):
    _fields_ = ([("conv_w", c_float_p * DC_NUM_VGG_CONVS), ("conv_b", c_float_p * DC_NUM_VGG_CONVS)] +
                [(n, c_float_p) for n in (
                    "rpn_conv_w", "rpn_conv_b", "rpn_box_w", "rpn_box_b", "rpn_score_w", "rpn_score_b",
                    "fc6_w", "fc6_b", "fc7_w", "fc7_b", "obj_w", "obj_b", "boxreg_w", "boxreg_b",
                    "lm_enc_w", "lm_enc_b", "lm_emb", "lstm_w", "lstm_b", "lm_out_w", "lm_out_b", "anchors")] +
                [("field_centers", C.c_float * 4)] +
                [(n, C.c_int32) for n in ("num_anchors", "rpn_hidden", "vocab_size", "seq_length", "enc_size",
                                          "rnn_size", "fc_dim")])


class DcResult(C.Structure):
    _fields_ = [("capacity", C.c_int32), ("K", C.c_int32), ("T", C.c_int32),
                ("boxes", c_float_p), ("scores", c_float_p), ("tokens", c_int32_p)]


class DenseCapError(RuntimeError):
    pass


_SIGS = {
    "dc_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int]),
    "dc_destroy": (None, [C.c_void_p]),
    "dc_last_error": (C.c_char_p, [C.c_void_p]),
    "dc_load_weights": (C.c_int, [C.c_void_p, C.POINTER(DcWeights)]),
    "dc_set_test_args": (C.c_int, [C.c_void_p, C.c_float, C.c_float, C.c_int]),
    "dc_set_localization_test_args": (C.c_int, [C.c_void_p, C.c_int, C.c_float, C.c_int]),
    "dc_set_lanes": (C.c_int, [C.c_void_p, C.c_int]),
    "dc_set_caption_order": (C.c_int, [C.c_void_p, C.c_int]),
    "dc_preprocess_size": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "dc_preprocess_u8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "dc_set_math_mode": (C.c_int, [C.c_void_p, C.c_int]),
    "dc_set_graph_replay": (C.c_int, [C.c_void_p, C.c_int]),
    "dc_set_beam_size": (C.c_int, [C.c_void_p, C.c_int]),
    "dc_set_group": (C.c_int, [C.c_void_p, C.c_int]),
    "dc_forward_test": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(DcResult)]),
    "dc_forward_batch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(DcResult)]),
    "dc_forward_images": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int,
                                    C.c_int, C.POINTER(DcResult)]),
    "dc_extract_features": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                      C.c_void_p, c_int32_p]),
    "dc_extract_features_images": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                             C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, c_int32_p]),
    "dc_stage_times": (C.c_int, [C.c_void_p, C.POINTER(C.c_char_p), c_float_p, C.c_int]),
    "dc_mfma_profile": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_double),
                                  C.POINTER(C.c_double)]),
    "dc_debug_fetch": (C.c_int64, [C.c_void_p, C.c_char_p, C.c_void_p, C.c_int64]),
    "dc_debug_set": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int64]),
    "dc_debug_plan_gemm": (C.c_int, [C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int, c_int32_p]),
    "dc_comm_unique_id": (C.c_int, [C.c_void_p]),
    "dc_comm_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p, C.c_int, C.c_int]),
    "dc_comm_create_ex": (C.c_int, [C.POINTER(C.c_void_p), C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "dc_comm_transport": (C.c_char_p, [C.c_void_p]),
    "dc_comm_destroy": (None, [C.c_void_p]),
    "dc_comm_last_error": (C.c_char_p, [C.c_void_p]),
    "dc_gather_results": (C.c_int, [C.c_void_p, C.POINTER(DcResult), C.c_int, C.POINTER(DcResult)]),
    "dc_malloc": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.c_size_t]),
    "dc_free": (C.c_int, [C.c_void_p, C.c_void_p]),
    "dc_memcpy_h2d": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "dc_memcpy_d2h": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "dc_synchronize": (C.c_int, [C.c_void_p]),
    "dc_op_chw_to_hwc": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "dc_op_hwc_to_chw": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "dc_op_pack_conv3x3_weights": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]),
    "dc_op_conv3x3": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                C.c_int, C.c_int, C.c_int, C.c_int]),
    "dc_op_conv3x3_relu_pool": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                          C.c_int, C.c_int]),
    "dc_op_conv3x3_c3": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                   C.c_int, C.c_int]),
    "dc_op_maxpool2x2_ceil": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]),
    "dc_op_linear": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                               C.c_int, C.c_int]),
    "dc_op_make_anchors": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float,
                                     C.c_float, C.c_void_p, C.c_int]),
    "dc_op_apply_box_transform": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "dc_op_clip_boxes": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_float,
                                   C.c_float, C.c_float]),
    "dc_op_xcycwh_to_x1y1x2y2": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "dc_op_box_iou": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "dc_op_rpn_decode": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_float,
                                   C.c_float, C.c_float, C.c_float, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "dc_op_nms": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_int,
                            C.c_void_p, C.c_void_p]),
    "dc_op_bilinear_roi_pool": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                          C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]),
    "dc_op_lm_sample": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
}

EXPORTED_SYMBOLS = tuple(_SIGS)
_lib = None


def lib():
    """Load libdensecap_hip.so (once) and attach the prototypes.  Raises if it is missing."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise DenseCapError(
                "libdensecap_hip.so not built at %s -- run `make -C densecap_amd/csrc` "
                "(there is no CPU fallback)" % LIB_PATH)
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGS.items():
            fn = getattr(l, name)  # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def check(ctx, rc, what=""):
    if rc < 0:
        msg = lib().dc_last_error(ctx)
        raise DenseCapError("%s failed (%d): %s" % (what or "densecap call", rc,
                                                    msg.decode() if msg else "?"))
    return rc
