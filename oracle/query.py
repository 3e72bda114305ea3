"""CPU oracle of `RSPrompterQuery.predict` (reference mmdet/rsprompter/models.py:173-272 with
RSMask2FormerHead :274-463,633-658, RSMaskFormerFusionHead :661-715).  TEST INFRASTRUCTURE ONLY.

Restated from the cited reference lines (mmdet) and, for the two mmcv bricks whose source is not under
/root/reference (`MultiScaleDeformableAttention`, `MultiheadAttention`, `FFN`), from their documented
semantics -- PARITY UNPINNED for those two.  The SAM pieces are HF modules.
The module tree reproduces the reference's `state_dict` keys.
"""
import math

import einops
import torch
import torch.nn.functional as F
from torch import nn

from . import glue, hf_sam
from .anchor import AnchorOracle, FeatureAggregator, SimpleFPN, _Wrap


# ----------------------------------------------------------------------------- mmcv bricks (App. B)
class ConvGN(nn.Module):
    """mmcv ConvModule(conv -> GroupNorm [-> ReLU]); norm sub-module is named `gn`."""

    def __init__(self, cin, cout, k, bias, act, groups=32):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, k, padding=k // 2, bias=bias)
        self.gn = nn.GroupNorm(groups, cout)
        self.act = act

    def forward(self, x):
        x = self.gn(self.conv(x))
        return F.relu(x) if self.act else x


class FFN(nn.Module):
    """mmcv FFN(num_fcs=2, ReLU): x + Linear(ReLU(Linear(x))); keys layers.0.0.*, layers.1.*"""

    def __init__(self, dim, hidden):
        super().__init__()
        self.layers = nn.Sequential(nn.Sequential(nn.Linear(dim, hidden), nn.ReLU(inplace=True)),
                                    nn.Linear(hidden, dim))

    def forward(self, x):
        return x + self.layers(x)


class MSDeformAttn(nn.Module):
    """mmcv MultiScaleDeformableAttention(embed_dims=128, num_heads=8, num_levels=3, num_points=4,
    batch_first=True) -- App. B: identity=query; query+=query_pos; value_proj; offsets/weights linears
    on the (pos-added) query; softmax over levels*points; loc = ref + offset / (W, H);
    grid_sample(bilinear, zeros, align_corners=False) on 2*loc-1; output_proj; + identity."""

    def __init__(self, dim=128, heads=8, levels=3, points=4):
        super().__init__()
        self.dim, self.heads, self.levels, self.points = dim, heads, levels, points
        self.sampling_offsets = nn.Linear(dim, heads * levels * points * 2)
        self.attention_weights = nn.Linear(dim, heads * levels * points)
        self.value_proj = nn.Linear(dim, dim)
        self.output_proj = nn.Linear(dim, dim)

    def forward(self, query, query_pos, reference_points, spatial_shapes):
        identity = query
        q = query + query_pos
        bs, nq, _ = q.shape
        H, L, P = self.heads, self.levels, self.points
        value = self.value_proj(query).view(bs, nq, H, -1)
        off = self.sampling_offsets(q).view(bs, nq, H, L, P, 2)
        w = self.attention_weights(q).view(bs, nq, H, L * P).softmax(-1).view(bs, nq, H, L, P)
        norm = torch.stack([spatial_shapes[..., 1], spatial_shapes[..., 0]], -1).to(q)     # (W, H) per level
        loc = reference_points[:, :, None, :, None, :] + off / norm[None, None, None, :, None, :]
        out = self._sample(value, spatial_shapes, loc, w)
        return self.output_proj(out) + identity

    @staticmethod
    def _sample(value, spatial_shapes, loc, w):
        bs, _, H, D = value.shape
        _, nq, _, L, P, _ = loc.shape
        vals = value.split([int(h * w_) for h, w_ in spatial_shapes], dim=1)
        grids = 2 * loc - 1
        samp = []
        for lvl, (h, w_) in enumerate(spatial_shapes):
            v = vals[lvl].flatten(2).transpose(1, 2).reshape(bs * H, D, int(h), int(w_))
            g = grids[:, :, :, lvl].transpose(1, 2).flatten(0, 1)                      # [bs*H, nq, P, 2]
            samp.append(F.grid_sample(v, g, mode='bilinear', padding_mode='zeros', align_corners=False))
        w = w.transpose(1, 2).reshape(bs * H, 1, nq, L * P)
        out = (torch.stack(samp, dim=-2).flatten(-2) * w).sum(-1).view(bs, H * D, nq)
        return out.transpose(1, 2).contiguous()


class MHA(nn.Module):
    """mmcv MultiheadAttention(batch_first=True) over nn.MultiheadAttention (`.attn`): q=query+query_pos,
    k=key+key_pos, v=value; bool attn_mask True = blocked; returns identity(query) + out."""

    def __init__(self, dim=128, heads=8):
        super().__init__()
        self.attn = nn.MultiheadAttention(dim, heads, dropout=0.0, batch_first=True)

    def forward(self, query, key, value, query_pos, key_pos, attn_mask=None):
        out = self.attn(query + query_pos, key + key_pos, value, attn_mask=attn_mask, need_weights=False)[0]
        return query + out


class EncLayer(nn.Module):
    """DeformableDetrTransformerEncoderLayer: detr_layers.py:213-238, deformable_detr_layers.py:237-249."""

    def __init__(self, dim=128, ffn=512, levels=3):
        super().__init__()
        self.self_attn = MSDeformAttn(dim, levels=levels)
        self.ffn = FFN(dim, ffn)
        self.norms = nn.ModuleList([nn.LayerNorm(dim), nn.LayerNorm(dim)])

    def forward(self, q, pos, ref, shapes):
        q = self.norms[0](self.self_attn(q, pos, ref, shapes))
        return self.norms[1](self.ffn(q))


class PixelDecoder(nn.Module):
    """MSDeformAttnPixelDecoder: msdeformattn_pixel_decoder.py:21-246 (5 inputs; `levels` = num_encoder_levels =
    self_attn_cfg.num_levels, 3 in every shipped config; `num_outs` memories are handed to the transformer decoder)."""

    def __init__(self, feat=128, out=256, strides=(4, 8, 16, 32, 64), ffn=512, enc_layers=3, levels=3, num_outs=3):
        super().__init__()
        self.strides, self.n_in, self.n_enc, self.pe_feats = list(strides), 5, levels, feat // 2
        self.num_outs = num_outs
        self.input_convs = nn.ModuleList([ConvGN(256, feat, 1, True, False) for _ in range(levels)])
        self.encoder = nn.Module()
        self.encoder.layers = nn.ModuleList([EncLayer(feat, ffn, levels) for _ in range(enc_layers)])
        self.level_encoding = nn.Embedding(levels, feat)
        self.lateral_convs = nn.ModuleList([ConvGN(256, feat, 1, False, False) for _ in range(5 - levels)])
        self.output_convs = nn.ModuleList([ConvGN(feat, feat, 3, False, True) for _ in range(5 - levels)])
        self.mask_feature = nn.Conv2d(feat, out, 1)

    def forward(self, feats):
        bs = feats[0].shape[0]
        inputs, poss, shapes, refs = [], [], [], []
        for i in range(self.n_enc):
            lvl = self.n_in - i - 1
            f = feats[lvl]
            h, w = f.shape[-2:]
            proj = self.input_convs[i](f)
            pe = glue.sine_positional_encoding(bs, h, w, num_feats=self.pe_feats)
            pos = self.level_encoding.weight[i].view(1, -1, 1, 1) + pe
            # MlvlPointGenerator(offset=0.5) / (W, H) * stride -> ((x+.5)/W, (y+.5)/H)   (:177-182)
            s = self.strides[lvl]
            sx = (torch.arange(0, w) + 0.5) * s
            sy = (torch.arange(0, h) + 0.5) * s
            xx = sx.repeat(h)
            yy = sy.view(-1, 1).repeat(1, w).view(-1)
            ref = torch.stack([xx, yy], -1) / (torch.tensor([w, h], dtype=torch.float32) * s)
            inputs.append(proj.flatten(2).permute(0, 2, 1))
            poss.append(pos.flatten(2).permute(0, 2, 1))
            shapes.append((h, w))
            refs.append(ref)
        q = torch.cat(inputs, 1)
        pos = torch.cat(poss, 1)
        ref = torch.cat(refs, 0)[None, :, None].repeat(bs, 1, self.n_enc, 1)
        sp = torch.tensor(shapes)
        for layer in self.encoder.layers:
            q = layer(q, pos, ref, sp)
        mem = q.permute(0, 2, 1)
        outs = list(torch.split(mem, [h * w for h, w in shapes], dim=-1))
        outs = [o.reshape(bs, -1, shapes[i][0], shapes[i][1]) for i, o in enumerate(outs)]
        for i in range(self.n_in - self.n_enc - 1, -1, -1):        # i = 1, 0 ; module index == feature index (:232-242)
            cur = self.lateral_convs[i](feats[i])
            y = cur + F.interpolate(outs[-1], size=cur.shape[-2:], mode='bilinear', align_corners=False)
            outs.append(self.output_convs[i](y))
        return self.mask_feature(outs[-1]), outs[:self.num_outs]


class DecLayer(nn.Module):
    """Mask2FormerTransformerDecoderLayer: mask2former_layers.py:73-135."""

    def __init__(self, dim=128, ffn=512):
        super().__init__()
        self.self_attn, self.cross_attn = MHA(dim), MHA(dim)
        self.ffn = FFN(dim, ffn)
        self.norms = nn.ModuleList([nn.LayerNorm(dim) for _ in range(3)])

    def forward(self, query, key, value, query_pos, key_pos, cross_attn_mask):
        query = self.norms[0](self.cross_attn(query, key, value, query_pos, key_pos, cross_attn_mask))
        query = self.norms[1](self.self_attn(query, query, query, query_pos, query_pos))
        return self.norms[2](self.ffn(query))


class QueryHead(nn.Module):
    """RSMask2FormerHead (decoder_plus=True): models.py:274-463."""

    def __init__(self, num_classes, num_queries, per_pointset_point=5, feat=128, out=256, decoder_plus=True,
                 with_sincos=True, input_proj=False, levels=3, multimask_output=False):
        """decoder_plus=False (models.py:303-307, 361-385): no mask-embedding MLP, `no_mask_embed` as the dense prompt, the
        SAM decoder runs in every stage and its masks drive the attention masks; with_sincos=False (models.py:315-318,
        346-347): the point MLP emits the prompts directly; input_proj: `enforce_decoder_input_project=True`
        (mask2former_head.py:93-100: Conv2d 1x1 per level); levels: `num_transformer_feat_level` ==
        the pixel decoder's `num_levels` (mask2former_head.py:106-107), with `num_outs` = levels memories.
        multimask_output=True (models.py:369-380): the decoder's three masks per prompt set are folded into the query axis
        by `mask_pred.reshape(img_bs, -1, h, w)` -> [B, 3 Nq, h, w], mask 3 q + j = mask j of prompt set q; the class
        predictions stay [B, Nq, .].  Only with decoder_plus=True: without it the folded masks are the attention-mask
        source and nn.MultiheadAttention rejects their shape (pinned in test_oracle_forwards.py)."""
        super().__init__()
        if multimask_output and not decoder_plus:
            raise ValueError('multimask_output=True needs decoder_plus=True (the reference fails in its first decoder layer)')
        self.multimask_output = bool(multimask_output)
        self.num_classes, self.num_queries, self.npts, self.num_heads = num_classes, num_queries, per_pointset_point, 8
        self.decoder_plus, self.with_sincos = decoder_plus, with_sincos
        self.levels = levels
        self.pixel_decoder = PixelDecoder(feat, out, levels=levels, num_outs=max(levels, 3))
        self.transformer_decoder = nn.Module()
        self.transformer_decoder.layers = nn.ModuleList([DecLayer() for _ in range(6)])
        self.transformer_decoder.post_norm = nn.LayerNorm(feat)
        self.query_embed = nn.Embedding(num_queries, feat)
        self.query_feat = nn.Embedding(num_queries, feat)
        self.level_embed = nn.Embedding(levels, feat)
        self.cls_embed = nn.Sequential(nn.Linear(feat, feat), nn.ReLU(inplace=True), nn.Linear(feat, num_classes + 1))
        if decoder_plus:
            self.mask_embed = nn.Sequential(nn.Linear(feat, feat), nn.ReLU(inplace=True), nn.Linear(feat, feat),
                                            nn.ReLU(inplace=True), nn.Linear(feat, out))
        self.point_emb = nn.Sequential(nn.Linear(feat, feat // 2), nn.ReLU(inplace=True),
                                       nn.Linear(feat // 2, feat // 2), nn.ReLU(inplace=True),
                                       nn.Linear(feat // 2, out * (2 if with_sincos else 1) * per_pointset_point))
        self.mask_decoder = _Wrap('mask_decoder', hf_sam.build_mask_decoder())
        if decoder_plus:
            self.sam_mask_embed = hf_sam.build_mask_embedding()
        else:
            self.no_mask_embed = nn.Embedding(1, out)
        if input_proj:
            self.decoder_input_projs = nn.ModuleList([nn.Conv2d(feat, feat, 1) for _ in range(levels)])
        self.input_proj = input_proj

    def _forward_head(self, decoder_out, mask_feature, attn_size, emb, ipe, run_sam):
        bs = emb.shape[0]
        decoder_out = self.transformer_decoder.post_norm(decoder_out)
        cls_pred = self.cls_embed(decoder_out)
        mask_pred_plus = torch.einsum('bqc,bchw->bqhw', self.mask_embed(decoder_out), mask_feature) if self.decoder_plus else None
        mask_pred, sparse = None, None
        if run_sam or not self.decoder_plus:     # models.py:644-646 keeps only the last call's SAM output
            pe = self.point_emb(decoder_out)
            pe = einops.rearrange(pe, 'b n_set (n_point c) -> b n_set n_point c', n_point=self.npts)
            if self.with_sincos:
                pe = torch.sin(pe[..., ::2]) + pe[..., 1::2]
            sparse = einops.rearrange(pe, 'b n_set n_point c -> (b n_set) n_point c').unsqueeze(1)
            if self.decoder_plus:
                dense = self.sam_mask_embed(einops.repeat(mask_pred_plus, 'b n h w -> (b n) c h w', c=1))
            else:
                # models.py:365: expand(img_bs, ...) only broadcasts for one image; the intent -- the same no-mask vector
                # for every prompt set -- restated for any batch
                he, we = emb.shape[-2:]
                dense = self.no_mask_embed.weight.reshape(1, -1, 1, 1).expand(bs * self.num_queries, -1, he, we)
            masks, _ = self.mask_decoder.mask_decoder(
                image_embeddings=torch.repeat_interleave(emb, self.num_queries, 0),
                image_positional_embeddings=torch.repeat_interleave(ipe, self.num_queries, 0),
                sparse_prompt_embeddings=sparse, dense_prompt_embeddings=dense, multimask_output=self.multimask_output)
            mask_pred = masks.reshape(bs, -1, *masks.shape[-2:])
        attn_src = mask_pred_plus if self.decoder_plus else mask_pred          # models.py:380-385
        attn_mask = F.interpolate(attn_src, attn_size, mode='bilinear', align_corners=False)
        attn_mask = attn_mask.flatten(2).unsqueeze(1).repeat((1, self.num_heads, 1, 1)).flatten(0, 1)
        attn_mask = attn_mask.sigmoid() < 0.5
        return cls_pred, mask_pred, attn_mask, mask_pred_plus, sparse

    @torch.no_grad()
    def forward(self, x, emb, ipe):
        bs = x[0].shape[0]
        mask_features, mem = self.pixel_decoder(x)
        dec_in, dec_pos = [], []
        for i in range(self.levels):
            mi = self.decoder_input_projs[i](mem[i]) if self.input_proj else mem[i]
            d = mi.flatten(2).permute(0, 2, 1) + self.level_embed.weight[i].view(1, 1, -1)
            pe = glue.sine_positional_encoding(bs, mem[i].shape[-2], mem[i].shape[-1], num_feats=64)
            dec_in.append(d)
            dec_pos.append(pe.flatten(2).permute(0, 2, 1))
        qf = self.query_feat.weight.unsqueeze(0).repeat((bs, 1, 1))
        qe = self.query_embed.weight.unsqueeze(0).repeat((bs, 1, 1))
        trace = dict(mask_features=mask_features, memory=mem, attn_masks=[], query_feats=[qf], levels=self.levels)
        src_of = lambda mpp_, mask_: mpp_ if self.decoder_plus else mask_           # what the attention masks are cut from
        cls, mask0, attn_mask, mpp, _ = self._forward_head(qf, mask_features, mem[0].shape[-2:], emb, ipe, False)
        trace.update(cls_pred_all=[cls], mask_pred_plus_all=[src_of(mpp, mask0)])
        for i in range(6):
            lvl = i % self.levels
            attn_mask = attn_mask & (attn_mask.sum(-1) != attn_mask.shape[-1]).unsqueeze(-1)   # models.py:439-442
            trace['attn_masks'].append(attn_mask)
            qf = self.transformer_decoder.layers[i](qf, dec_in[lvl], dec_in[lvl], qe, dec_pos[lvl], attn_mask)
            trace['query_feats'].append(qf)
            cls, mask, attn_mask, mpp, sparse = self._forward_head(
                qf, mask_features, mem[(i + 1) % self.levels].shape[-2:], emb, ipe, run_sam=(i == 5))
            trace['cls_pred_all'].append(cls)
            trace['mask_pred_plus_all'].append(src_of(mpp, mask))
        trace.update(cls_pred=cls, mask_pred=mask, mask_pred_plus=mpp, sparse_embeddings=sparse)
        return cls, mask, trace


def mask2bbox(masks):
    """structures/mask/utils.py:56-77."""
    n = masks.shape[0]
    bboxes = masks.new_zeros((n, 4), dtype=torch.float32)
    x_any, y_any = torch.any(masks, dim=1), torch.any(masks, dim=2)
    for i in range(n):
        x, y = torch.where(x_any[i, :])[0], torch.where(y_any[i, :])[0]
        if len(x) > 0 and len(y) > 0:
            bboxes[i, :] = bboxes.new_tensor([x[0], y[0], x[-1] + 1, y[-1] + 1])
    return bboxes


def instance_postprocess(mask_cls, mask_pred, num_classes, max_per_image=100):
    """maskformer_fusion_head.py:126-182; topk(sorted=False) order is unspecified in the reference:
    canonical (score desc, flat index asc)."""
    scores = F.softmax(mask_cls, dim=-1)[:, :-1]
    nq = mask_cls.shape[0]
    labels = torch.arange(num_classes).unsqueeze(0).repeat(nq, 1).flatten(0, 1)
    flat = scores.flatten(0, 1)
    k = min(max_per_image, flat.numel())
    ranked, order = flat.sort(descending=True, stable=True)
    scores_per_image, top = ranked[:k], order[:k]
    labels_per_image = labels[top]
    query_indices = top // num_classes
    mask_pred = mask_pred[query_indices]
    binary = (mask_pred > 0).float()
    mask_scores = (mask_pred.sigmoid() * binary).flatten(1).sum(1) / (binary.flatten(1).sum(1) + 1e-6)
    det_scores = scores_per_image * mask_scores
    binary = binary.bool()
    return dict(bboxes=mask2bbox(binary), labels=labels_per_image, scores=det_scores, masks=binary,
                query_indices=query_indices, cls_scores=scores_per_image, mask_logits=mask_pred)


def fusion_predict(mask_cls_results, mask_pred_results, metas, num_classes, max_per_image=100, rescale=True):
    """RSMaskFormerFusionHead.predict (models.py:663-715) with instance_on only: crop the padding
    (`int(ori * scale_factor)`), optionally resize the LOGITS to ori_shape, then instance_postprocess."""
    results = []
    for c, m, meta in zip(mask_cls_results, mask_pred_results, metas):
        oh, ow = meta['ori_shape'][:2]
        sf = meta['scale_factor']
        m = m[:, :int(oh * sf[1]), :int(ow * sf[0])]
        if rescale:
            m = F.interpolate(m[:, None], size=(oh, ow), mode='bilinear', align_corners=False)[:, 0]
        results.append(instance_postprocess(c, m, num_classes, max_per_image))
    return results


class QueryOracle(nn.Module):
    """RSPrompterQuery predict path, configs/rsprompter/_base_/rsprompter_query.py."""

    def __init__(self, arch='base', num_classes=1, num_queries=100, select_layers=None, max_per_image=100, lora=None,
                 peft512=False, head_kwargs=None):
        """lora=dict(r, alpha): RSSamVisionEncoder(peft_config=...) (models.py:785-797; BASELINE.json configs[4]);
        peft512=True: the rsprompter_query-nwpu-peft-512.py tree (ViTSAM at 512 px + LoRA + PseudoFeatureAggregator)."""
        super().__init__()
        depth = hf_sam.ARCH[arch]['num_hidden_layers']
        select_layers = list(select_layers) if select_layers is not None else list(range(1, depth + 1, 2))
        self.num_classes, self.max_per_image, self.peft512 = num_classes, max_per_image, peft512
        self.shared_image_embedding = _Wrap('shared_image_embedding', hf_sam.build_positional_embedding(arch))
        self.neck = nn.Module()
        if peft512:
            from .anchor import PseudoAggregator
            from .vitsam import PeftWrapped, ViTSAM
            self.backbone = _Wrap('vision_encoder', PeftWrapped(ViTSAM(arch, 512, lora=True)))
            self.neck.feature_aggregator = PseudoAggregator(256, 512, 256)
        else:
            self.backbone = _Wrap('vision_encoder', hf_sam.build_vision_encoder(arch, lora=lora))
            self.neck.feature_aggregator = FeatureAggregator(arch, 32, 256, select_layers)
        self.neck.feature_spliter = SimpleFPN()
        self.panoptic_head = QueryHead(num_classes, num_queries, **(head_kwargs or {}))
        self.eval()

    extract_feat = AnchorOracle.extract_feat

    @torch.no_grad()
    def predict(self, batch_inputs, metas, rescale=True):
        x, emb, ipe, t0 = self.extract_feat(batch_inputs)
        cls, mask, trace = self.panoptic_head(x, emb, ipe)
        img_shape = metas[0]['batch_input_shape']
        mask_up = F.interpolate(mask, size=(img_shape[0], img_shape[1]), mode='bilinear', align_corners=False)
        results = fusion_predict(cls, mask_up, metas, self.num_classes, self.max_per_image, rescale)
        trace.update(t0, fpn=x, image_embeddings=emb, image_pe=ipe)
        return results, trace
