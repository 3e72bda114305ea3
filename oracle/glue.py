"""Plain-PyTorch restatement of the mmdet glue on RSPrompter's anchor predict path
(oracle; tests only).  Every function names the reference lines it follows.
Tie-breaking wherever the reference leaves it to the sort implementation is the
canonical "score descending, original index ascending" (SURVEY.md App. D.10).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

from . import cops


# --------------------------------------------------------------------------- anchors
def gen_base_anchors(base_size, scales, ratios, center_offset=0.0):
    """anchor_generator.py:161-205 (scale_major=True, center=None)."""
    w = h = float(base_size)
    x_c, y_c = center_offset * w, center_offset * h
    ratios = torch.tensor(ratios, dtype=torch.float32)
    scales = torch.tensor(scales, dtype=torch.float32)
    h_ratios = torch.sqrt(ratios)
    w_ratios = 1 / h_ratios
    ws = (w * w_ratios[:, None] * scales[None, :]).view(-1)
    hs = (h * h_ratios[:, None] * scales[None, :]).view(-1)
    return torch.stack([x_c - 0.5 * ws, y_c - 0.5 * hs, x_c + 0.5 * ws, y_c + 0.5 * hs], dim=-1)


def grid_priors(featmap_sizes, strides, scales, ratios):
    """anchor_generator.py:230-301: per level [H*W*A, 4], position-major then anchor."""
    out = []
    for (fh, fw), s in zip(featmap_sizes, strides):
        base = gen_base_anchors(s, scales, ratios)
        sx = torch.arange(0, fw, dtype=torch.float32) * s
        sy = torch.arange(0, fh, dtype=torch.float32) * s
        xx = sx.repeat(fh)
        yy = sy.view(-1, 1).repeat(1, fw).view(-1)
        shifts = torch.stack([xx, yy, xx, yy], dim=-1)
        out.append((base[None, :, :] + shifts[:, None, :]).view(-1, 4))
    return out


# --------------------------------------------------------------------------- box coder
def delta2bbox(rois, deltas, means=(0., 0., 0., 0.), stds=(1., 1., 1., 1.), max_shape=None,
               wh_ratio_clip=16 / 1000, clip_border=True, add_ctr_clamp=False, ctr_clamp=32):
    """delta_xywh_bbox_coder.py:264-361."""
    num_bboxes, num_classes = deltas.size(0), deltas.size(1) // 4
    if num_bboxes == 0:
        return deltas
    deltas = deltas.reshape(-1, 4)
    means = deltas.new_tensor(means).view(1, -1)
    stds = deltas.new_tensor(stds).view(1, -1)
    d = deltas * stds + means
    dxy, dwh = d[:, :2], d[:, 2:]
    rois_ = rois.repeat(1, num_classes).reshape(-1, 4)
    pxy = (rois_[:, :2] + rois_[:, 2:]) * 0.5
    pwh = rois_[:, 2:] - rois_[:, :2]
    dxy_wh = pwh * dxy
    max_ratio = np.abs(np.log(wh_ratio_clip))
    if add_ctr_clamp:                                   # :340-342
        dxy_wh = torch.clamp(dxy_wh, max=ctr_clamp, min=-ctr_clamp)
        dwh = torch.clamp(dwh, max=max_ratio)
    else:
        dwh = dwh.clamp(min=-max_ratio, max=max_ratio)
    gxy = pxy + dxy_wh
    gwh = pwh * dwh.exp()
    x1y1 = gxy - gwh * 0.5
    x2y2 = gxy + gwh * 0.5
    bboxes = torch.cat([x1y1, x2y2], dim=-1)
    if clip_border and max_shape is not None:
        bboxes[..., 0::2].clamp_(min=0, max=max_shape[1])
        bboxes[..., 1::2].clamp_(min=0, max=max_shape[0])
    return bboxes.reshape(num_bboxes, -1)


# --------------------------------------------------------------------------- nms
def batched_nms(boxes, scores, idxs, iou_threshold, split_thr=10000):
    """mmcv.ops.batched_nms (un-vendored): coordinate-offset trick in fp32,
    single nms below split_thr, per-id loop + global re-sort otherwise."""
    if boxes.numel() == 0:
        return torch.cat([boxes, scores[:, None]], -1), torch.zeros((0,), dtype=torch.long)
    max_coordinate = boxes.max()
    offsets = idxs.to(boxes) * (max_coordinate + torch.tensor(1).to(boxes))
    boxes_for_nms = boxes + offsets[:, None]
    if boxes_for_nms.shape[0] < split_thr:
        dets, keep = cops.nms(boxes_for_nms, scores, iou_threshold)
        boxes = boxes[keep]
        scores = dets[:, -1]
    else:
        total_mask = scores.new_zeros(scores.size(), dtype=torch.bool)
        for i in torch.unique(idxs):
            mask = (idxs == i).nonzero(as_tuple=False).view(-1)
            _, keep = cops.nms(boxes_for_nms[mask], scores[mask], iou_threshold)
            total_mask[mask[keep]] = True
        keep = total_mask.nonzero(as_tuple=False).view(-1)
        scores, inds = scores[keep].sort(descending=True, stable=True)
        keep = keep[inds]
        boxes = boxes[keep]
    return torch.cat([boxes, scores[:, None]], -1), keep


def multiclass_nms(multi_bboxes, multi_scores, score_thr, iou_threshold, max_num):
    """bbox_nms.py:12-105.  returns dets [k,5], labels [k], flat candidate index [k]."""
    num_classes = multi_scores.size(1) - 1
    bboxes = multi_bboxes.view(multi_scores.size(0), -1, 4)
    scores = multi_scores[:, :-1]
    labels = torch.arange(num_classes, dtype=torch.long).view(1, -1).expand_as(scores)
    bboxes = bboxes.reshape(-1, 4)
    scores = scores.reshape(-1)
    labels = labels.reshape(-1)
    valid = scores > score_thr
    inds = valid.nonzero(as_tuple=False).squeeze(1)
    bboxes, scores, labels = bboxes[inds], scores[inds], labels[inds]
    if bboxes.numel() == 0:
        return torch.cat([bboxes, scores[:, None]], -1), labels, inds
    dets, keep = batched_nms(bboxes, scores, labels, iou_threshold)
    if max_num > 0:
        dets, keep = dets[:max_num], keep[:max_num]
    return dets, labels[keep], inds[keep]


def bbox_head_predict_single(roi, cls_score, bbox_pred, img_shape, num_classes, score_thr, iou_threshold, max_per_img,
                             stds=(0.1, 0.1, 0.2, 0.2), scale_factor=None, coder=None):
    """BBoxHead._predict_by_feat_single (bbox_head.py:476-571), class-specific regression (scale_factor=None: rescale=False):
    softmax scores, per-class delta decode of the repeated RoIs, multiclass NMS.  roi [n,5], cls_score [n,nc+1],
    bbox_pred [n,nc*4] -> dets [k,5], labels [k], flat (roi, class) candidate index [k]."""
    n = roi.shape[0]
    scores = torch.softmax(cls_score, dim=-1)
    # `coder`: the remaining delta2bbox keywords of the head's DeltaXYWHBBoxCoder (means, clip_border, add_ctr_clamp, ...)
    bboxes = delta2bbox(roi[:, 1:].repeat_interleave(num_classes, dim=0), bbox_pred.view(-1, 4),
                        max_shape=img_shape, **dict(dict(stds=stds), **(coder or {})))
    if scale_factor is not None and bboxes.size(0) > 0:
        # rescale=True (bbox_head.py:549-552 + scale_boxes, structures/bbox/transforms.py:391-414): a python reciprocal,
        # then an fp32 product, before the NMS
        inv = [1 / s for s in scale_factor]
        bboxes = bboxes * bboxes.new_tensor(inv).repeat((1, int(bboxes.size(-1) / 2)))
    bboxes = bboxes.view(n, -1)
    return multiclass_nms(bboxes, scores, score_thr, iou_threshold, max_per_img)


# --------------------------------------------------------------------------- RPN
def rpn_predict_single(cls_score_list, bbox_pred_list, mlvl_priors, img_shape, nms_pre=1000,
                       max_per_img=1000, iou_thr=0.7, min_bbox_size=0, coder=None, use_sigmoid_cls=True):
    """rpn_head.py:134-304 for one image; cls/bbox lists are [A*1,H,W] / [A*4,H,W] (softmax objectness,
    use_sigmoid_cls=False: [A*2,H,W] as [fg, bg] per anchor, :193-200)."""
    mlvl_bbox, mlvl_prior, mlvl_score, level_ids, mlvl_src = [], [], [], [], []
    for lvl, (cls, reg, priors) in enumerate(zip(cls_score_list, bbox_pred_list, mlvl_priors)):
        reg = reg.permute(1, 2, 0).reshape(-1, 4)
        if use_sigmoid_cls:
            scores = cls.permute(1, 2, 0).reshape(-1, 1).sigmoid().squeeze(1)
        else:
            scores = cls.permute(1, 2, 0).reshape(-1, 2).softmax(-1)[:, :-1].squeeze(1)
        src = torch.arange(scores.shape[0])
        if 0 < nms_pre < scores.shape[0]:
            ranked, rank_inds = scores.sort(descending=True, stable=True)
            topk = rank_inds[:nms_pre]
            scores = ranked[:nms_pre]
            reg, priors, src = reg[topk], priors[topk], topk
        mlvl_bbox.append(reg)
        mlvl_prior.append(priors)
        mlvl_score.append(scores)
        mlvl_src.append(src)
        level_ids.append(scores.new_full((scores.size(0),), lvl, dtype=torch.long))
    reg = torch.cat(mlvl_bbox)
    priors = torch.cat(mlvl_prior)
    bboxes = delta2bbox(priors, reg, max_shape=img_shape, **(coder or {}))
    scores = torch.cat(mlvl_score)
    level_ids = torch.cat(level_ids)
    src = torch.cat(mlvl_src)
    if min_bbox_size >= 0:
        w = bboxes[:, 2] - bboxes[:, 0]
        h = bboxes[:, 3] - bboxes[:, 1]
        valid = (w > min_bbox_size) & (h > min_bbox_size)
        if not valid.all():
            bboxes, scores, level_ids, src = bboxes[valid], scores[valid], level_ids[valid], src[valid]
    if bboxes.numel() > 0:
        dets, keep = batched_nms(bboxes, scores, level_ids, iou_thr)
        keep = keep[:max_per_img]
        return dict(bboxes=bboxes[keep], scores=dets[:max_per_img, -1], level_ids=level_ids[keep],
                    anchor_index=src[keep])
    return dict(bboxes=bboxes.new_zeros((0, 4)), scores=scores.new_zeros(0),
                level_ids=level_ids.new_zeros(0), anchor_index=src.new_zeros(0))


# --------------------------------------------------------------------------- RoI extractor
def map_roi_levels(rois, num_levels, finest_scale=56):
    """single_level_roi_extractor.py:44-63."""
    scale = torch.sqrt((rois[:, 3] - rois[:, 1]) * (rois[:, 4] - rois[:, 2]))
    lvls = torch.floor(torch.log2(scale / finest_scale + 1e-6))
    return lvls.clamp(min=0, max=num_levels - 1).long()


def roi_extract(feats, rois, out_size, strides, finest_scale=56):
    """single_level_roi_extractor.py:65-119 with mmcv RoIAlign(sampling_ratio=0, aligned=True)."""
    n = len(strides)
    out = feats[0].new_zeros(rois.size(0), feats[0].shape[1], out_size, out_size)
    lvls = map_roi_levels(rois, n, finest_scale)
    for i in range(n):
        inds = (lvls == i).nonzero(as_tuple=False).squeeze(1)
        if inds.numel() > 0:
            out[inds] = cops.roi_align(feats[i], rois[inds], out_size, 1.0 / strides[i], 0, True)
    return out


# --------------------------------------------------------------------------- positional encodings
def sine_positional_encoding(B, H, W, num_feats=128, temperature=10000, scale=2 * math.pi, eps=1e-6):
    """positional_encoding.py:60-110 with an all-valid mask, normalize=True, offset=0."""
    not_mask = torch.ones((B, H, W), dtype=torch.int)
    y_embed = not_mask.cumsum(1, dtype=torch.float32)
    x_embed = not_mask.cumsum(2, dtype=torch.float32)
    y_embed = y_embed / (y_embed[:, -1:, :] + eps) * scale
    x_embed = x_embed / (x_embed[:, :, -1:] + eps) * scale
    dim_t = torch.arange(num_feats, dtype=torch.float32)
    dim_t = temperature ** (2 * (dim_t // 2) / num_feats)
    pos_x = x_embed[:, :, :, None] / dim_t
    pos_y = y_embed[:, :, :, None] / dim_t
    pos_x = torch.stack((pos_x[:, :, :, 0::2].sin(), pos_x[:, :, :, 1::2].cos()), dim=4).view(B, H, W, -1)
    pos_y = torch.stack((pos_y[:, :, :, 0::2].sin(), pos_y[:, :, :, 1::2].cos()), dim=4).view(B, H, W, -1)
    return torch.cat((pos_y, pos_x), dim=3).permute(0, 3, 1, 2)


def image_wide_positional_embeddings(pos_matrix, size):
    """models.py:85-95 + HF:552-566: grid (i+0.5)/size -> 2x-1 -> @G -> 2pi -> [sin, cos]."""
    grid = torch.ones((size, size), dtype=pos_matrix.dtype)
    y_embed = (grid.cumsum(dim=0) - 0.5) / size
    x_embed = (grid.cumsum(dim=1) - 0.5) / size
    coords = torch.stack([x_embed, y_embed], dim=-1)
    coords = 2 * coords - 1
    coords = coords @ pos_matrix
    coords = 2 * np.pi * coords
    pe = torch.cat([torch.sin(coords), torch.cos(coords)], dim=-1)
    return pe.permute(2, 0, 1).unsqueeze(0)


# --------------------------------------------------------------------------- mask post-process
def mask_postprocess_single(mask_preds, bboxes, img_meta, mask_thr_binary=0.5, rescale=True):
    """models.py:1746-1784 (activate_map=False).  mask_preds [k,1,h,w] logits.  returns
    (bool masks [k,H,W], bboxes possibly rescaled in place, the fp32 probability map)."""
    scale_factor = bboxes.new_tensor(img_meta['scale_factor']).repeat((1, 2))
    img_h, img_w = img_meta['ori_shape'][:2]
    mask_preds = mask_preds.sigmoid()
    if rescale:
        bboxes = bboxes / scale_factor
    else:
        w_scale, h_scale = scale_factor[0, 0], scale_factor[0, 1]
        img_h = np.round(img_h * h_scale.item()).astype(np.int32)
        img_w = np.round(img_w * w_scale.item()).astype(np.int32)
    im_mask = F.interpolate(mask_preds, size=img_meta['batch_input_shape'], mode='bilinear',
                            align_corners=False).squeeze(1)
    scale_factor_w, scale_factor_h = img_meta['scale_factor']
    ori_rescaled_size = (img_h * scale_factor_h, img_w * scale_factor_w)
    im_mask = im_mask[:, :int(ori_rescaled_size[0]), :int(ori_rescaled_size[1])]
    h, w = img_meta['ori_shape'][:2]
    prob = F.interpolate(im_mask.unsqueeze(1), size=(h, w), mode='bilinear', align_corners=False).squeeze(1)
    return prob >= mask_thr_binary, bboxes, prob


# --------------------------------------------------------------------------- data preprocessor
def data_preprocess(imgs, mean, std, bgr_to_rgb=True, pad_size_divisor=32, pad_value=0):
    """data_preprocessor.py:110-149 + mmengine ImgDataPreprocessor: list of [3,H,W] -> [B,3,Hp,Wp]."""
    mean = torch.tensor(mean, dtype=torch.float32).view(-1, 1, 1)
    std = torch.tensor(std, dtype=torch.float32).view(-1, 1, 1)
    outs = []
    for im in imgs:
        im = im.float()
        if bgr_to_rgb:
            im = im[[2, 1, 0], ...]
        outs.append((im - mean) / std)
    hm = max(o.shape[1] for o in outs)
    wm = max(o.shape[2] for o in outs)
    hp = int(math.ceil(hm / pad_size_divisor)) * pad_size_divisor
    wp = int(math.ceil(wm / pad_size_divisor)) * pad_size_divisor
    batch = torch.full((len(outs), 3, hp, wp), float(pad_value))
    for i, o in enumerate(outs):
        batch[i, :, :o.shape[1], :o.shape[2]] = o
    return batch
