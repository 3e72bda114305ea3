"""ctypes binding of librsp_hip.so (the C ABI declared in include/rsp_hip.h).

The product path has NO fallback: if the library is missing, or a call returns
a non-zero status, a RuntimeError is raised.  Nothing here imports `oracle/`.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "librsp_hip.so")

c_void_p = ctypes.c_void_p
c_int = ctypes.c_int
c_int64 = ctypes.c_int64
c_float = ctypes.c_float


class RspGemmDesc(ctypes.Structure):
    _fields_ = [
        ("A", c_void_p), ("Bhi", c_void_p), ("Blo", c_void_p), ("C", c_void_p),
        ("bias", c_void_p), ("res", c_void_p), ("a_rowmap", c_void_p), ("c_rowmap", c_void_p),
        ("M", c_int), ("N", c_int), ("K", c_int),
        ("lda", c_int), ("ldc", c_int), ("ldr", c_int),
        ("res_mod", c_int), ("act", c_int),
        ("alpha", c_float), ("a_scale_log2", c_int),
        ("conv_k", c_int), ("conv_stride", c_int), ("conv_pad", c_int),
        ("conv_H", c_int), ("conv_W", c_int), ("conv_C", c_int),
        ("conv_Ho", c_int), ("conv_Wo", c_int),
        ("ct_W", c_int), ("ct_dy", c_int),
        ("res_bmap", c_void_p), ("res_brows", c_int),
        ("Ahi", c_void_p), ("Alo", c_void_p), ("Chi", c_void_p), ("Clo", c_void_p), ("c_scale_log2", c_int),
        ("a_rows", c_int), ("c_rows", c_int), ("b_rows", c_int),
        ("ln_gamma", c_void_p), ("ln_beta", c_void_p), ("ln_eps", c_float),
        ("res_hi", c_void_p), ("res_lo", c_void_p), ("res_scale_log2", c_int), ("res_rows", c_int),
        ("hd_hyper", c_void_p), ("hd_out", c_void_p), ("hd_rows", c_int),
        ("c_ncols", c_int), ("pl_col0", c_int),
        ("tile_hint", c_int),
    ]


class RspAttnDesc(ctypes.Structure):
    _fields_ = [
        ("q", c_void_p), ("k", c_void_p), ("v", c_void_p), ("out", c_void_p), ("kv_batch_map", c_void_p),
        ("q_batch_map", c_void_p),
        ("q_bs", c_int64), ("q_ts", c_int64), ("q_hs", c_int64),
        ("k_bs", c_int64), ("k_ts", c_int64), ("k_hs", c_int64),
        ("v_bs", c_int64), ("v_ts", c_int64), ("v_hs", c_int64),
        ("o_bs", c_int64), ("o_ts", c_int64), ("o_hs", c_int64),
        ("B", c_int), ("nh", c_int), ("dh", c_int), ("Tq", c_int), ("Tk", c_int),
        ("scale", c_float),
        ("out_hi", c_void_p), ("out_lo", c_void_p), ("out_scale_log2", c_int),
        ("mask", c_void_p),
    ]


class RspMaskEmbedDesc(ctypes.Structure):
    _fields_ = [(n, c_void_p) for n in (
        "mask_pred_plus", "image_embeddings", "roi_img", "conv1_w", "conv1_b", "ln1_w", "ln1_b", "conv2_w",
        "conv2_b", "ln2_w", "ln2_b", "conv3_w", "conv3_b", "out")] + [
        ("R", c_int), ("he", c_int), ("we", c_int), ("C", c_int), ("eps", c_float)]


class RspI2tFusedDesc(ctypes.Structure):
    _fields_ = [("q", c_void_p), ("q_map", c_void_p), ("k", c_void_p), ("v", c_void_p), ("wo", c_void_p), ("bo", c_void_p),
                ("res", c_void_p), ("res_map", c_void_p), ("res_hi", c_void_p), ("res_lo", c_void_p),
                ("res_scale_log2", c_int), ("gamma", c_void_p), ("beta", c_void_p), ("eps", c_float),
                ("out", c_void_p), ("out_hi", c_void_p), ("out_lo", c_void_p), ("out_scale_log2", c_int),
                ("R", c_int), ("T", c_int), ("N", c_int), ("scale", c_float)]


class RspRoiAlignDesc(ctypes.Structure):
    _fields_ = [
        ("feat", c_void_p * 4), ("pe", c_void_p * 4), ("H", c_int * 4), ("W", c_int * 4),
        ("spatial_scale", c_float * 4), ("rois", c_void_p), ("out", c_void_p),
        ("K", c_int), ("P", c_int), ("C", c_int), ("num_levels", c_int), ("finest_scale", c_int),
    ]


class RspBoxCoder(ctypes.Structure):
    _fields_ = [("means", c_float * 4), ("stds", c_float * 4), ("max_ratio", c_float), ("ctr_clamp", c_float),
                ("clip_border", c_int), ("add_ctr_clamp", c_int)]


class RspRpnDesc(ctypes.Structure):
    _fields_ = [
        ("head", c_void_p * 5), ("H", c_int * 5), ("W", c_int * 5), ("stride", c_float * 5),
        ("ld", c_int), ("A", c_int), ("nms_pre", c_int), ("num_levels", c_int),
        ("base_anchors", c_void_p), ("coder", RspBoxCoder), ("min_bbox_size", c_float),
    ]


# name -> (restype, argtypes); the CPU test checks every symbol is exported.
PROTOTYPES = {
    "rsp_abi_version": (c_int, []),
    "rsp_build_info": (ctypes.c_char_p, []),
    "rsp_split_f16": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "rsp_split_f16_kb32": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p]),
    "rsp_gemm": (c_int, [ctypes.POINTER(RspGemmDesc), c_void_p]),
    "rsp_layernorm": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_float, c_int, c_void_p]),
    "rsp_layernorm_ex": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int64, c_int,
                                 c_float, c_int, c_void_p]),
    "rsp_vit_attention_ex": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                     c_int, c_float, c_void_p]),
    "rsp_vit_relpos": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "rsp_vit_attention_global_ws_bytes": (ctypes.c_int64, [c_int, c_int, c_int, c_int]),
    "rsp_vit_attention_global": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                         c_int, c_int, c_float, c_void_p]),
    "rsp_vit_attention_planes": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p,
                                         c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p]),
    "rsp_vit_attention_planes_ex": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p,
                                            c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_float, c_int, c_int,
                                            c_void_p]),
    "rsp_vit_window_attention": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p,
                                         c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_int, c_int, c_int,
                                         c_void_p]),
    "rsp_pack_relpos_tables": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "rsp_vit_relpos_rows": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p,
                                    c_int64, c_void_p]),
    "rsp_vit_relpos_q": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "rsp_vit_attention": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_float, c_void_p]),
    "rsp_preprocess": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int,
                               ctypes.POINTER(c_float), ctypes.POINTER(c_float), c_int, c_float, c_void_p]),
    "rsp_paste_masks": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p,
                                c_void_p]),
    "rsp_resize_pad": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int,
                               ctypes.POINTER(c_float), c_int, c_int, ctypes.POINTER(c_float), ctypes.POINTER(c_float),
                               c_void_p]),
    "rsp_patchify": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "rsp_attention": (c_int, [ctypes.POINTER(RspAttnDesc), c_void_p]),
    "rsp_mask_rle": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "rsp_sam_upscale2": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                                 c_void_p, c_int, c_int, c_void_p]),
    "rsp_sam_t2i_attention": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_void_p]),
    "rsp_sam_i2t_fused": (c_int, [ctypes.POINTER(RspI2tFusedDesc), c_void_p]),
    "rsp_sam_i2t_attention": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                                      c_int, c_int, c_int, c_float, c_void_p]),
    "rsp_roi_align": (c_int, [ctypes.POINTER(RspRoiAlignDesc), c_void_p]),
    "rsp_rpn_topk": (c_int, [ctypes.POINTER(RspRpnDesc), c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "rsp_rpn_decode": (c_int, [ctypes.POINTER(RspRpnDesc), c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int,
                               c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "rsp_bbox_post": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float,
                              ctypes.POINTER(RspBoxCoder), c_int, c_void_p, c_void_p, c_void_p, c_void_p,
                              c_void_p, c_void_p]),
    "rsp_sam_t2i_fold": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int,
                                 c_void_p, c_void_p, c_int, c_int64, c_void_p, c_int, c_int, c_int, c_void_p]),
    "rsp_sam_fold_expand": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_float, c_void_p]),
    "rsp_sam_fold_gather": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "rsp_sam_upscale_fused": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p,
                                      c_float, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int,
                                      c_void_p]),
    "rsp_nms_workspace_bytes": (c_int64, [c_int, c_int]),
    "rsp_batched_nms": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_float, c_int,
                                c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "rsp_hyper_mask": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "rsp_mask_post": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_float,
                              c_void_p, c_void_p, c_void_p]),
    "rsp_pool2": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "rsp_add_rows": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p]),
    "rsp_sincos_pairs": (c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
    "rsp_div_boxes": (c_int, [c_void_p, c_void_p, c_int64, ctypes.POINTER(c_float), c_void_p]),
    "rsp_fill_bias_rows": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p, c_void_p, c_int64,
                                   c_int, c_int, c_void_p]),
    "rsp_scale_boxes": (c_int, [c_void_p, c_void_p, c_int64, ctypes.POINTER(c_float), c_void_p]),
    "rsp_pack_bits": (c_int, [c_void_p, c_void_p, c_int64, c_void_p]),
    "rsp_groupnorm_nhwc": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                   c_int, c_float, c_int, c_void_p]),
    "rsp_resize_bilinear_nhwc": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "rsp_msdeform_attn": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int,
                                  ctypes.POINTER(c_int), c_void_p]),
    "rsp_msdeform_attn_ex": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int,
                                     ctypes.POINTER(c_int), c_int, c_void_p]),
    "rsp_query_attn_mask": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_int, c_void_p]),
    "rsp_sam_mask_embed": (c_int, [ctypes.POINTER(RspMaskEmbedDesc), c_void_p]),
    "rsp_query_topk": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "rsp_query_mask_post": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                    c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "rsp_gather_rows": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "rsp_gemm_uses_s2": (c_int, [ctypes.POINTER(RspGemmDesc)]),
    "rsp_gemm_s2_epilogue": (c_int, [ctypes.POINTER(RspGemmDesc)]),
    "rsp_gemm_uses_pp": (c_int, [ctypes.POINTER(RspGemmDesc)]),
    "rsp_rle_to_string": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int64, c_void_p]),
    "rsp_mask_post_logits": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_float,
                                     c_void_p, c_void_p, c_void_p]),
    "rsp_resnet_stem": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "rsp_maxpool_nhwc": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "rsp_upsample_nearest_add": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "rsp_sam_embed_boxes": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int,
                                    c_void_p]),
}

_lib = None


def load():
    """Load librsp_hip.so (after torch, so both share one HIP runtime)."""
    global _lib
    if _lib is not None:
        return _lib
    import torch  # noqa: F401  -- makes libamdhip64.so.7 resident first
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `python -m rsprompter_amd.build` "
            "(hipcc --offload-arch=gfx950). There is no fallback path.")
    # a library left over from older sources would be loaded silently and its struct layouts (RspGemmDesc, ...) would
    # no longer match this file: compare the source digest recorded at build time; rebuild when hipcc is here, else fail
    from . import build as _build
    try:
        stale = (not os.path.exists(_build.STAMP)) or open(_build.STAMP).read().strip() != _build._digest()
    except OSError:
        stale = True
    if stale:
        if os.path.exists(_build.HIPCC) and os.access(os.path.dirname(LIB_PATH), os.W_OK):
            _build.build(verbose=False)
        else:
            raise RuntimeError(f"{LIB_PATH} was built from different sources than the ones in this tree "
                               "(digest mismatch) and hipcc is not available to rebuild it")
    lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(status, what):
    if status != 0:
        raise RuntimeError(f"librsp_hip: {what} failed with status {status}")
