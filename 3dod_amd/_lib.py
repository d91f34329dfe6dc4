"""ctypes binding of libcr3dod.so (include/cr3dod.h).  The product path FAILS
LOUDLY when the library is missing -- there is no CPU / eager fallback."""
import ctypes
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libcr3dod.so")

c_void_p, c_int, c_int64, c_float = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float
P = c_void_p

# name -> argtypes; the single source of truth for "every symbol the header declares"
SIGNATURES = {
    "cr_ctx_create": [c_int, P, ctypes.POINTER(P)],
    "cr_ctx_destroy": [P],
    "cr_ctx_set_stream": [P, P],
    "cr_abi_version": [],
    "cr_cuboid_corners": [P, P, P, c_int64, P],
    "cr_cubes_project_score": [P, P, c_int64, c_int64, P, c_int, c_float, c_float, P, P, P, P,
                               P, P, P, P, P, P, P, P, P],
    "cr_cubes_project_score_fast": [P, P, c_int64, c_int64, P, c_int, c_float, c_float, P, P, P, P,
                                    P, P, P, P, P, P, P, P, P, P],
    "cr_propose": [P, P, c_int64, P, c_int, c_int, P, P, P, c_int64, P, c_int, P, P, P, P, P],
    "cr_ransac_plane": [P, P, c_int64, P, c_int64, c_float, P, P, P],
    "cr_ransac_plane_batched": [P, P, P, c_int, c_int64, P, c_int64, c_float, P, P, P],
    "cr_propose_batched": [P, P, P, c_int64, P, c_int, c_int, c_int, P, P, P, c_int64, P, c_int, P, P, P, P, P],
    "cr_box_median": [P, P, c_int, c_int, c_int, P, P, c_int, P],
    "cr_fold_bn": [P, P, P, P, P, P, c_float, P, P, c_int, c_int, c_int],
    "cr_hull8": [P, P, c_int, P, P, P],
    "cr_segment_counts": [P, P, c_int, P, c_int, c_int, c_int, P],
    "cr_mask_rects": [P, P, P, c_int, c_int, c_int, P, P, P, P, P, P],
    "cr_polygon_focal": [P, P, P, P, P, P, c_int, c_int, c_int, P, P],
    "cr_attention_fwd": [P, P, P, c_int, c_int, c_int, c_int, c_float],
    "cr_layernorm": [P, P, P, P, P, c_int64, c_int, c_float],
    "cr_gelu_inplace": [P, P, c_int64],
    "cr_scale_residual_layernorm": [P, P, P, P, P, P, P, P, c_int64, c_int, c_float],
    "cr_scale_residual": [P, P, P, P, P, c_int64, c_int],
    "cr_resize_bilinear_ac": [P, P, P, c_int, c_int, c_int, c_int, c_int, c_int],
    "cr_conv2d_fwd": [P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, P, c_int, P, c_int, c_int, P],
    "cr_conv2d_bwd_data": [P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, P],
    "cr_weight_split3": [P, P, P, c_int64, c_int],
    "cr_relu_bwd": [P, P, P, P, c_int64, c_int],
    "cr_gt_pack": [P, P, P, P, P, P, c_int, c_int, P, P, P, P],
    "cr_rpn_unpack": [P, P, P, c_int, c_int, c_int, c_int, P, P, P],
    "cr_rpn_pack_grad": [P, P, P, P, P, c_int, c_int, c_int, c_int],
    "cr_topk_blocks": [c_int64, c_int],
    "cr_topk": [P, P, c_int, c_int64, c_int, P, P, P],
    "cr_multi_seg": [P, P, c_int, c_int64, c_int],
    "cr_loss_guard": [P, P, c_int, c_float, P, P, P, c_int, c_float, c_float, P],
    "cr_step_counters": [P, P, P, P],
    "cr_weights_split3": [P, P, P, P, c_int, c_int64],
    "cr_conv2d_bwd_weight": [P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int],
    "cr_conv2d_bwd_weight_bias": [P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int],
    "cr_cast_f32_to_bf16": [P, P, P, c_int64],
    "cr_weight_transpose": [P, P, P, c_int, c_int, c_int, c_int],
    "cr_colsum_accum": [P, P, c_int, c_int64, c_int, P, P],
    "cr_bn_fwd": [P, P, P, c_int, P, P, P, P, c_int64, c_int, c_int, c_float, c_float, P, P, P, c_int],
    "cr_bn_bwd": [P, P, P, P, P, P, P, P, P, P, P, c_int64, c_int, c_int, c_int],
    "cr_pool2x_fwd": [P, P, P, c_int, c_int, c_int, c_int, c_int, c_int],
    "cr_pool2x_bwd": [P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P],
    "cr_upsample2x_add": [P, P, P, P, c_int, c_int, c_int, c_int, c_int],
    "cr_sum2x2": [P, P, P, c_int, c_int, c_int, c_int, c_int],
    "cr_preprocess": [P, P, P, c_int, c_int, c_int, P, P, c_int],
    "cr_roi_align_fwd": [P, P, P, P, P, c_int, c_int, P, c_int64, c_int, c_int, P, c_int],
    "cr_roi_align_bwd": [P, P, P, P, P, c_int, c_int, P, c_int64, c_int, c_int, P, c_int],
    "cr_conv2d_fwd_group": [P, c_int, P, P, P, P, P, P, c_int, c_int, c_int, c_int, P, P, c_int, c_int],
    "cr_wino_filter": [P, P, P, c_int, c_int, c_int],
    "cr_gemm_batched_f32": [P, P, P, P, c_int, c_int, c_int, c_int, c_int64, c_int64, c_int64],
    "cr_wino_input": [P, c_int, P, P, P, P, c_int, P, c_int64],
    "cr_wino_output": [P, c_int, P, P, P, P, P, c_int, c_int64, P, c_int, P],
    "cr_wino_dy": [P, c_int, P, P, P, P, c_int, P, c_int64, P],
    "cr_wino_filter_grad": [P, P, P, c_int, c_int, P, P],
    "cr_wgrad_batched_f32": [P, P, P, P, c_int, c_int, c_int, c_int, c_int64, c_int64, c_int64],
    "cr_conv2d_bwd_data_group": [P, c_int, P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, P],
    "cr_conv2d_bwd_weight_group": [P, c_int, P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int],
    "cr_roi_align_bwd_set": [P, P, P, P, P, c_int, c_int, c_int, P, c_int64, c_int, c_int, P, c_int],
    "cr_nms_grouped": [P, P, P, c_int, c_int, c_float, P, P],
    "cr_cube_loss_fwd": [P, P, c_int64, c_int, c_int, c_int, c_int, P, P],
    "cr_cube_loss_bwd": [P, P, c_int64, c_int, c_int, c_int, c_int, P, P, P, P, P, P],
    "cr_rpn_decode_select": [P, P, P, P, P, c_int, c_int, c_int, P, c_float, P, c_float, P, P, P],
    "cr_box_match": [P, P, c_int, P, P, c_int, c_int, c_int, P, P, P, P],
    "cr_rpn_label": [P, P, P, P, P, P, P, c_int, c_int, c_int, c_float, c_float, P, c_float, P, P, P, P],
    "cr_rpn_scatter": [P, P, P, c_int, P, P, c_int, c_int, P, c_float, c_int, c_int, P],
    "cr_rpn_loss": [P, P, P, P, P, P, P, c_int, c_int, c_int, P, P, P, P, P],
    "cr_roi_label": [P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_float, c_float, c_float, P, P, P],
    "cr_roi_compact": [P, P, P, c_int, P, P, c_int, c_int, P, P, P, c_int, c_int, P, P, P, P, P],
    "cr_box_loss": [P, P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, P, c_float, P, P, P, P, P],
    "cr_det_scores": [P, P, c_int, P, c_int, P, P, c_int, c_int, c_int, c_int, c_float, P, P],
    "cr_det_gather": [P, P, P, P, c_int, P, P, c_int, c_int, c_int, c_int, c_int, P, c_float, P, P, P, P],
    "cr_nms_grouped_cls": [P, P, P, P, c_int, c_int, c_float, P, P],
    "cr_det_pick": [P, P, P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P, P, P, P, P, P],
    "cr_weak_loss_fwd": [P, P, P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P, P, P, P, P],
    "cr_weak_loss_reduce": [P, P, P, P, P, c_int, c_int, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P, P, P, P, P, P, P, P,
                            P, P, P],
    "cr_weak_loss_bwd": [P, P, P, P, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P, P, P, P, P, P, P, P, P, P, P, P],
    "cr_cube_select": [P, P, c_int, P, c_int, P, P, P, c_int, c_int, c_int, c_int, P, P, P, P, P, P, P, c_int, c_int, P, P, P],
    "cr_cube_select_bwd": [P, P, c_int, P, c_int, c_int, c_int, P, P, P, P, P, P, P, P, P, c_int, c_int, P, P, P],
    "cr_cube_reduce": [P, P, P, P, P, c_int, c_int, P, P, P],
    "cr_cube_reduce_bwd": [P, P, P, P, c_int, c_int, P, P, P, P],
    "cr_weights_prepare": [P, P, P, P, P, P, c_int, c_int],
    "cr_fc_weight_prepare": [P, P, P, c_int, c_int, c_int, c_int],
    "cr_fc_grad_accum": [P, P, P, c_int, c_int, c_int, c_int],
    "cr_linear_fwd": [P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P],
    "cr_linear_bwd_data": [P, P, P, P, c_int, c_int, c_int, c_int, P],
    "cr_linear_bwd_weight": [P, P, P, P, P, c_int, c_int, c_int, c_int, c_int],
    "cr_transpose2d": [P, P, P, c_int, c_int, c_int],
    "cr_maxpool3x3s2_fwd": [P, P, P, c_int, c_int, c_int, c_int, c_int],
    "cr_maxpool3x3s2_bwd": [P, P, P, P, c_int, c_int, c_int, c_int, c_int],
    "cr_cube_decode_infer": [P, P, c_int, P, c_int, P, P, P, P, P, c_int, c_int, P, c_int, c_int, P, P],
    "cr_box3d_overlap": [P, P, P, c_int, c_int, P, P],
    "cr_nonfinite_flag": [P, P, c_int64, P],
    "cr_sgd_step": [P, P, P, P, c_int64, c_float, P, c_float, c_float, c_float, P],
    "cr_sgd_step_nesterov": [P, P, P, P, c_int64, c_float, P, c_float, c_float, c_float, P],
    "cr_grad_clip_value": [P, P, c_int64, c_float, c_float],
    "cr_grad_clip_norm": [P, P, P, P, c_int, c_float, c_float, c_float, P],
    "cr_adam_tick": [P, P, P],
    "cr_adam_step": [P, P, P, P, P, P, c_int64, c_float, P, c_float, c_float, c_float, c_float, c_float, c_int, P, P],
}


class CrError(RuntimeError):
    pass


_lib = None
_lock = threading.Lock()


def load():
    """dlopen libcr3dod.so and set signatures.  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise CrError(
                f"{LIB_PATH} not found: build it with `python 3dod_amd/build.py` "
                "(or __graft_entry__.build()). There is no fallback path.")
        # torch ships its own libamdhip64; it has to be in the process BEFORE this library pulls in a HIP runtime, or
        # two runtimes coexist and the second one sees no device ("no ROCm-capable device is detected")
        import torch  # noqa: F401
        lib = ctypes.CDLL(LIB_PATH)
        lib.cr_last_error.restype = ctypes.c_char_p
        lib.cr_last_error.argtypes = []
        for name, argtypes in SIGNATURES.items():
            fn = getattr(lib, name)          # AttributeError if the symbol is missing
            fn.argtypes = argtypes
            fn.restype = c_int
        _lib = lib
    return _lib


def check(rc, what):
    if rc != 0:
        msg = load().cr_last_error().decode("utf-8", "replace")
        raise CrError(f"{what} failed (rc={rc}): {msg}")


# ---- per-device context bound to torch's current stream ---------------------
_ctxs = {}


def ctx_for(device):
    """cr_ctx for a torch cuda device, re-pointed at torch's CURRENT stream on
    every call so library launches order with surrounding torch work."""
    import torch
    lib = load()
    if device.type != "cuda":
        raise CrError(f"3dod_amd kernels run on the GPU only (got device '{device}'); there is no CPU path")
    idx = device.index if device.index is not None else torch.cuda.current_device()
    stream = torch.cuda.current_stream(idx).cuda_stream
    ctx = _ctxs.get(idx)
    if ctx is None:
        h = P()
        check(lib.cr_ctx_create(idx, P(stream), ctypes.byref(h)), "cr_ctx_create")
        ctx = _ctxs[idx] = h
    else:
        check(lib.cr_ctx_set_stream(ctx, P(stream)), "cr_ctx_set_stream")
    return ctx


def ptr(t):
    """device pointer of a contiguous tensor, or NULL for None."""
    if t is None:
        return P(None)
    if not t.is_contiguous():
        import torch
        if not (t.dim() == 4 and t.is_contiguous(memory_format=torch.channels_last)):
            raise CrError("tensor must be dense (contiguous or channels_last)")
    return P(t.data_ptr())
