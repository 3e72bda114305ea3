"""CPU restatement of the reference's TEST PIPELINE front end (tests only):
    Resize(scale, keep_ratio=True) -> Pad(size, pad_val) -> PackDetInputs
configs/rsprompter/_base_/rsprompter_anchor.py:231-241; mmdet/datasets/transforms/transforms.py:134-247 (Resize over
mmcv.transforms.Resize._resize_img -> mmcv.imrescale -> cv2.resize(INTER_LINEAR)) and :704-786 (Pad over
mmcv.transforms.Pad._pad_img -> mmcv.impad(constant)); formatting.py PackDetInputs (HWC -> CHW tensor, meta keys).

mmcv and OpenCV are not under /root/reference and not installed: PARITY UNPINNED at that boundary.  The arithmetic is
restated from their documented behaviour -- mmcv 2.1 `rescale_size` / `_scale_size` (new = int(old * sf + 0.5)) and
OpenCV's `resizeGeneric_` for CV_32F INTER_LINEAR (source coordinate (d + 0.5) * (src / dst) - 0.5 with a double
scale, floor, clamp of the left/top tap to [0, size - 1], fp32 taps (1 - f, f), horizontal pass then vertical pass).
`test_cv2_linear_restatement_matches_torch_interpolate` cross-checks it against torch's independent implementation of
the same sampling rule (F.interpolate(bilinear, align_corners=False)), which agrees to fp32 rounding.
"""
import numpy as np


def rescale_size(old_wh, scale):
    w, h = old_wh
    sf = min(max(scale) / max(h, w), min(scale) / min(h, w))
    return (int(w * float(sf) + 0.5), int(h * float(sf) + 0.5)), sf


def _taps(dst, src):
    scale = float(src) / float(dst)                       # double
    f = ((np.arange(dst, dtype=np.float64) + 0.5) * scale - 0.5).astype(np.float32)
    i0 = np.floor(f).astype(np.int64)
    f = f - i0.astype(np.float32)
    lo = i0 < 0
    i0[lo], f[lo] = 0, 0.0
    hi = i0 >= src - 1
    i0[hi], f[hi] = src - 1, 0.0
    i1 = np.minimum(i0 + 1, src - 1)
    return i0, i1, f.astype(np.float32)


def cv2_resize_linear_f32(img, new_w, new_h):
    """img: [H, W, C] float32 -> [new_h, new_w, C] float32."""
    img = np.asarray(img, dtype=np.float32)
    H, W = img.shape[:2]
    x0, x1, fx = _taps(new_w, W)
    y0, y1, fy = _taps(new_h, H)
    fx = fx[None, :, None]
    rows0 = img[y0][:, x0] * (np.float32(1) - fx) + img[y0][:, x1] * fx          # horizontal pass on the top rows
    rows1 = img[y1][:, x0] * (np.float32(1) - fx) + img[y1][:, x1] * fx
    fy = fy[:, None, None]
    return (rows0 * (np.float32(1) - fy) + rows1 * fy).astype(np.float32)


def run_test_pipeline(img_bgr, scale=(1024, 1024), pad_size=(1024, 1024),
                      pad_val=(0.406 * 255, 0.456 * 255, 0.485 * 255)):
    """img_bgr: decoded [H, W, 3] uint8 (or float) array -> (inputs float32 [3, Hp, Wp], meta dict)."""
    img = np.asarray(img_bgr).astype(np.float32)           # LoadImageFromFile(to_float32=True)
    h, w = img.shape[:2]
    (nw, nh), _ = rescale_size((w, h), scale)
    res = cv2_resize_linear_f32(img, nw, nh)
    pw, ph = max(pad_size[0], nw), max(pad_size[1], nh)
    out = np.empty((ph, pw, 3), dtype=np.float32)
    out[...] = np.asarray(pad_val, dtype=np.float32)
    out[:nh, :nw] = res
    meta = dict(ori_shape=(h, w), img_shape=(ph, pw), scale_factor=(nw / w, nh / h), pad_shape=(ph, pw, 3))
    return np.ascontiguousarray(out.transpose(2, 0, 1)), meta
