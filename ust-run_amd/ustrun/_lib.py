"""ctypes binding of libustrun.so (include/ustrun.h).  There is no fallback: if the library is
missing or a call fails, a RuntimeError is raised."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("USTRUN_LIB", os.path.join(_HERE, "libustrun.so"))      # (USTRUN_LIB: A/B runs of two builds on one box)

F32, BF16, F16, F32X3 = 0, 1, 2, 3
LOSS_SOFTMAX, LOSS_SIGMOID = 0, 1

vp, fp, i32, i64, f32 = C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_float


class Src(C.Structure):
    """ustrun_src_t"""
    _fields_ = [("ptr", vp), ("scale", vp), ("shift", vp), ("C", i32), ("H", i32), ("W", i32),
                ("sN", i64), ("sH", i64), ("sW", i64), ("sC", i64),
                ("relu", i32), ("pool", i32), ("off_y", i32), ("off_x", i32), ("f32", i32),
                ("gN", i32), ("gstride", i64)]


class UNetDesc(C.Structure):
    """ustrun_unet_desc_t"""
    _fields_ = [("N", i32), ("C", i32), ("H", i32), ("W", i32), ("K", i32), ("base", i32), ("dtype", i32),
                ("train", i32), ("update_running", i32), ("groups", i32), ("tail", i32), ("lead", i32), ("momentum", f32), ("eps", f32),
                ("conv_w", vp * 18), ("bn_w", vp * 18), ("bn_b", vp * 18), ("bn_rm", vp * 18),
                ("bn_rv", vp * 18), ("bn_nbt", vp * 18), ("up_w", vp * 4), ("up_b", vp * 4),
                ("head_w", vp), ("head_b", vp), ("packed", vp), ("bilinear", i32)]


class AsmRow(C.Structure):
    """ustrun_asm_row_t"""
    _fields_ = [("a", vp), ("b", vp), ("box", vp)]


ASM_MAX, BBOX_BLOCKS = 128, 64


class ProfRec(C.Structure):
    """ustrun_prof_rec_t"""
    _fields_ = [("kind", i32), ("tag", i32), ("n", i32), ("pad", i32), ("ms", C.c_double), ("flops", C.c_double),
                ("bytes", C.c_double)]


PSrc, PDesc = C.POINTER(Src), C.POINTER(UNetDesc)

# name: (restype, argtypes) -- must list every symbol include/ustrun.h declares
SIGNATURES = {
    "ustrun_version": (i32, []),
    "ustrun_last_error": (C.c_char_p, []),
    "ustrun_pack_conv3x3": (i32, [fp, i32, i32, vp, vp, i32, vp]),
    "ustrun_pack_convT2x2": (i32, [fp, i32, i32, vp, vp, i32, vp]),
    "ustrun_conv_mtiles": (i32, [i32, i32, i32, i32]),
    "ustrun_conv3x3_fwd": (i32, [PSrc, i32, vp, i32, i32, i32, i32, vp, fp, i32, vp]),
    "ustrun_conv3x3_fwd_rows": (i32, [PSrc, i32, vp, i32, i32, i32, i32, vp, fp, C.POINTER(C.c_int), i32, vp]),
    "ustrun_bn_finalize": (i32, [fp, i32, i32, i64, fp, fp, fp, fp, vp, f32, f32, i32, fp, fp, fp, fp, vp]),
    "ustrun_bn_eval_affine": (i32, [i32, fp, fp, fp, fp, f32, fp, fp, vp]),
    "ustrun_bn_relu_apply": (i32, [vp, fp, fp, i64, i32, i32, fp, i32, i32, vp]),
    "ustrun_pool_act": (i32, [PSrc, i32, vp, i32, vp]),
    "ustrun_pool_act2": (i32, [PSrc, i32, vp, vp, i32, vp]),
    "ustrun_act16": (i32, [PSrc, i32, vp, i32, vp]),
    "ustrun_maxpool_bwd": (i32, [vp, vp, i32, i32, i32, i32, vp, i32, vp]),
    "ustrun_convT2x2_fwd": (i32, [PSrc, vp, fp, i32, i32, i32, i32, vp, i32, vp]),
    "ustrun_head_fwd": (i32, [vp, fp, fp, i64, i32, i32, i32, fp, fp, fp, i32, vp]),
    "ustrun_head_bwd": (i32, [fp, vp, fp, fp, i64, i32, i32, i32, fp, vp, fp, fp, i32, fp, i64, i32, vp]),
    "ustrun_bn_bwd_partials_bytes": (i64, [i64, i32]),
    "ustrun_bn_bwd_reduce": (i32, [vp, vp, vp, fp, fp, fp, fp, fp, i32, i32, i32, i32, fp, fp, i32, fp, fp, i64, i32, vp]),
    "ustrun_bn_bwd_apply": (i32, [vp, vp, vp, fp, fp, fp, i32, i32, i32, i32, vp, i32, vp]),
    "ustrun_conv3x3_dgrad": (i32, [vp, vp, i32, i32, i32, i32, i32, vp, i32, vp, i32, i32, i32, i32, i32, vp]),
    "ustrun_conv3x3_dgrad_bnsum": (i32, [vp, vp, i32, i32, i32, i32, i32, vp, vp, fp, fp, i32, i64, fp, C.POINTER(i32), i32, vp]),
    "ustrun_bn_bwd_finalize_stat": (i32, [fp, i32, i32, i32, i64, fp, fp, fp, i64, fp, fp, i32, fp, vp]),
    "ustrun_convT2x2_dgrad_bnsum": (i32, [vp, vp, i32, i32, i32, i32, i32, vp, vp, fp, fp, i32, i64, fp, C.POINTER(i32), i32, vp]),
    "ustrun_wgrad_partials_bytes": (i64, [i32, i32, i32, i64]),
    "ustrun_conv3x3_wgrad": (i32, [PSrc, i32, vp, i32, i32, i32, i32, fp, i32, fp, i64, i32, vp]),
    "ustrun_convT2x2_dgrad": (i32, [vp, vp, i32, i32, i32, i32, i32, vp, i32, vp]),
    "ustrun_convT2x2_wgrad": (i32, [PSrc, vp, i32, i32, i32, i32, fp, fp, i32, fp, i64, i32, vp]),
    "ustrun_pseudo_label": (i32, [fp, i32, i32, i32, f32, i32, vp, fp, vp]),
    "ustrun_mix_targets": (i32, [i32, i32, i32, i32, fp, vp, fp, vp, fp, vp, fp, vp, fp, vp, fp, vp, fp, vp, fp, vp]),
    "ustrun_box_mix": (i32, [fp, fp, fp, i32, i32, i32, fp, vp]),
    "ustrun_assemble": (i32, [vp, i32, i64, i32, vp, vp]),
    "ustrun_decode_labels": (i32, [fp, i32, i32, i32, vp, vp]),
    "ustrun_region_bbox": (i32, [vp, i32, i32, i32, i32, vp, vp]),
    "ustrun_upsample2x_fwd": (i32, [fp, i32, i32, i32, i32, fp, vp]),
    "ustrun_upsample2x_bwd": (i32, [fp, i32, i32, i32, i32, fp, vp]),
    "ustrun_rect_masks": (i32, [vp, i32, i32, i32, fp, vp]),
    "ustrun_upload_small": (i32, [vp, vp, i32, vp]),
    "ustrun_freq_mix_work_bytes": (i64, [i32, i32, i32]),
    "ustrun_freq_mix": (i32, [fp, fp, fp, i32, i32, i32, i32, i32, fp, vp, i64, vp]),
    "ustrun_loss_partials_bytes": (i64, [i32, i32, i32]),
    "ustrun_seg_loss_fwd": (i32, [fp, vp, fp, i32, i32, i32, i32, fp, fp, i64, vp]),
    "ustrun_seg_loss_bwd": (i32, [fp, vp, fp, i32, i32, i32, i32, fp, fp, f32, f32, f32, fp, vp]),
    "ustrun_dice_counts": (i32, [vp, vp, i32, i32, i32, i32, i32, i32, vp, vp]),
    "ustrun_sgd_ema": (i32, [fp, fp, fp, fp, i64, f32, f32, f32, i32, f32, f32, vp]),
    "ustrun_amp_check": (i32, [fp, i64, fp, vp]),
    "ustrun_sgd_ema_scaled": (i32, [fp, fp, fp, fp, i64, f32, f32, f32, i32, f32, f32, fp, vp]),
    "ustrun_amp_update": (i32, [fp, f32, f32, i32, vp]),
    "ustrun_pack_conv": (i32, [fp, i32, i32, i32, vp, i32, vp]),
    "ustrun_pack_conv_elems": (i64, [i32, i32, i32]),
    "ustrun_conv2d_fwd": (i32, [PSrc, i32, vp, fp, i32, i32, i32, i32, i32, i32, i32, vp, i32, fp, C.POINTER(C.c_int), i32, vp]),
    "ustrun_conv_rowwin_fwd": (i32, [PSrc, vp, i32, i32, i32, i32, i32, i32, vp, fp, C.POINTER(C.c_int), i32, vp]),
    "ustrun_maxpool3x3s2": (i32, [vp, fp, fp, i32, i32, i32, i32, vp, i32, vp]),
    "ustrun_bn_add_relu": (i32, [vp, fp, fp, vp, fp, fp, i64, i32, vp, i32, vp]),
    "ustrun_aspp_gather": (i32, [fp, i32, i32, i32, i32, i32, C.POINTER(C.c_int), fp, fp, vp]),
    "ustrun_sum_resize_bilinear": (i32, [C.POINTER(vp), i32, i32, i32, i32, i32, i32, i32, fp, vp]),
    "ustrun_rowwin_patches": (i32, [PSrc, i32, i32, i32, i32, i32, i32, vp, i32, vp]),
    "ustrun_relu_bwd_add": (i32, [vp, vp, vp, i64, vp, i32, vp]),
    "ustrun_upsample2x_act": (i32, [PSrc, i32, vp, i32, vp]),
    "ustrun_upsample2x_bwd_t": (i32, [vp, i32, i32, i32, i32, vp, i32, vp]),
    "ustrun_space_to_batch": (i32, [PSrc, i32, i32, vp, i32, vp]),
    "ustrun_conv2d_dgrad_bnsum": (i32, [vp, vp, i32, i32, i32, i32, i32, i32, vp, vp, fp, fp, fp, C.POINTER(C.c_int), i32, vp]),
    "ustrun_conv1x1_dgrad_join": (i32, [vp, vp, i32, i32, i32, i32, i32, vp, vp, vp, vp, fp, fp, fp, C.POINTER(C.c_int), C.POINTER(C.c_int), i32, vp]),
    "ustrun_maxpool3x3s2_bwd": (i32, [vp, vp, fp, fp, i32, i32, i32, i32, vp, i32, vp]),
    "ustrun_sum_resize_bilinear_bwd": (i32, [fp, i32, i32, i32, i32, i32, i32, fp, vp]),
    "ustrun_aspp_scatter": (i32, [fp, i32, i32, i32, i32, i32, C.POINTER(C.c_int), i32, vp, i32, vp]),
    "ustrun_colsum": (i32, [fp, i64, i32, fp, i32, vp]),
    "ustrun_conv2d_wgrad": (i32, [PSrc, i32, vp, i32, i32, i32, i32, i32, i32, i32, fp, i32, fp, i64, i32, vp]),
    "ustrun_conv_rowwin_wgrad": (i32, [PSrc, vp, i32, i32, i32, i32, i32, i32, fp, i32, fp, i64, i32, vp]),
    "ustrun_debug_last_conv_variant": (i32, []),
    "ustrun_dice_fwd": (i32, [fp, vp, i32, i32, fp, i32, i32, i32, i32, i32, i32, C.POINTER(C.c_float), fp, fp, i64, vp]),
    "ustrun_dice_bwd": (i32, [fp, vp, i32, i32, fp, i32, i32, i32, i32, i32, i32, C.POINTER(C.c_float), fp, fp, f32, fp, vp]),
    "ustrun_debug_conv_stat_rows": (i32, [i32] * 10),
    "ustrun_debug_last_wgrad_variant": (i32, []),
    "ustrun_debug_flags": (i32, [i32]),
    "ustrun_debug_flags2": (i32, [i32]),
    "ustrun_short_last_pass": (i32, [i32]),
    "ustrun_debug_buffer": (i32, [vp, i64]),
    "ustrun_debug_last_bn_variant": (i32, []),
    "ustrun_debug_clock_probe": (i32, [vp, i32, vp]),
    "ustrun_profile_enable": (i32, [i32]),
    "ustrun_profile_collect": (i32, [i32, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(i64)]),
    "ustrun_profile_stream": (i32, [vp, i32]),
    "ustrun_profile_records": (i64, [C.POINTER(ProfRec), i64]),
    "ustrun_unet_packed_bytes": (i64, [PDesc]),
    "ustrun_unet_fwd_workspace_bytes": (i64, [PDesc]),
    "ustrun_unet_bwd_scratch_bytes": (i64, [PDesc]),
    "ustrun_unet_pack": (i32, [PDesc, vp]),
    "ustrun_unet_forward": (i32, [PDesc, fp, fp, fp, vp, vp]),
    "ustrun_unet_backward": (i32, [PDesc, fp, fp, vp, vp, C.POINTER(vp), i32, vp]),
    "ustrun_unet_backward_part": (i32, [PDesc, fp, fp, vp, vp, C.POINTER(vp), i32, i32, vp]),
}

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is not built. Run `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C ust-run_amd/csrc`. There is no CPU fallback for this path.")
        # torch first: its wheel carries the HIP runtime the device memory and streams come from, and libustrun.so must bind to THAT
        # copy -- loaded on its own it pulls in /opt/rocm's, and a process with both sees "no ROCm-capable device" at the first launch
        import torch  # noqa: F401
        h = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(h, name)          # AttributeError if the library lacks a declared symbol
            fn.restype, fn.argtypes = res, args
        _lib = h
    return _lib


def check(rc, what="ustrun"):
    if rc != 0:
        msg = lib().ustrun_last_error()
        raise RuntimeError(f"{what} failed ({rc}): {msg.decode() if msg else '?'}")


def ptr(t):
    """device/host pointer of a tensor, or None"""
    return None if t is None else t.data_ptr()


def nhwc_src(t_ptr, C_, H, W, scale=None, shift=None, relu=0, pool=0, off=(0, 0), f32=0, gN=0, gstride=0):
    """gN > 0: image n takes scale/shift + (n // gN) * gstride floats (several forward passes batched into one call)."""
    return Src(t_ptr, scale, shift, C_, H, W, H * W * C_, W * C_, C_, 1, relu, pool, off[0], off[1], f32, gN, gstride)


def nchw_src(t_ptr, C_, H, W):
    """The network input: NCHW, always f32."""
    return Src(t_ptr, None, None, C_, H, W, C_ * H * W, W, 1, H * W, 0, 0, 0, 0, 1)
