"""Golden vectors for the SAMSeg sibling model's mask branch, from the REAL reference file
mmdet/models/roi_heads/mask_heads/fcn_mask_head.py (stubs / stand-ins of make_golden_forwards.py):
  FCNMaskHead.{__init__, forward}            :27-150   (on the ConvModule stand-in; upsample / predictor = torch layers)
  FCNMaskHead._predict_by_feat_single        :276-420
  _do_paste_mask                             :423-480
and for SAMSegMask2Former's head the REAL mmdet/models/dense_heads/mask2former_head.py (Mask2FormerHead.__init__ :62-156,
_forward_head :340-380, forward :382-460) over msdeformattn_pixel_decoder.py / mask2former_layers.py.
python tests/golden/make_golden_samseg.py -> tests/golden/reference_vectors_samseg.pt"""
import os
import sys
import types

import torch
from torch import nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import make_golden as mg  # noqa: E402
import make_golden_forwards as mf  # noqa: E402

OUT = os.path.join(HERE, 'reference_vectors_samseg.pt')


@torch.no_grad()
def main():
    reg, st = mf.install()
    mods = sys.modules

    def build_upsample_layer(cfg):
        cfg = dict(cfg)
        assert cfg.pop('type') == 'deconv'
        return nn.ConvTranspose2d(**cfg)

    def build_conv_layer(cfg, *a, **k):
        assert cfg is None or cfg.get('type') in ('Conv', 'Conv2d')
        return nn.Conv2d(*a, **k)

    mods['mmcv.cnn'].build_upsample_layer = build_upsample_layer
    mods['mmcv.cnn'].build_conv_layer = build_conv_layer
    mods['mmcv.ops.carafe'] = types.ModuleType('mmcv.ops.carafe')
    mods['mmcv.ops.carafe'].CARAFEPack = type('CARAFEPack', (nn.Module,), {})
    mods['mmengine.config'].ConfigDict = mf.CD
    fcn = mg._load('mmdet/models/roi_heads/mask_heads/fcn_mask_head.py', '')
    out = {}
    head = mf.seeded(fcn.FCNMaskHead(num_convs=4, in_channels=256, conv_out_channels=256, num_classes=10), 31)
    x = mf.rnd(301, 3, 256, 14, 14)
    out['fcn_head'] = dict(keys=mf.keyshapes(head), seed=31, x=(301, (3, 256, 14, 14)), out=head(x).clone())
    # paste + predict_single on smooth logits, several metas (rescale on / off, non-square, boxes over the border)
    cases = []
    for seed, meta, rescale in ((0, dict(ori_shape=(64, 96), scale_factor=(1.0, 1.0)), True),
                                (1, dict(ori_shape=(40, 60), scale_factor=(2.0, 2.0)), True),
                                (2, dict(ori_shape=(50, 30), scale_factor=(1.5, 1.5)), False)):
        g = torch.Generator().manual_seed(700 + seed)
        n, nc = 6, 4
        logits = torch.nn.functional.avg_pool2d(torch.randn(n, nc, 28, 28, generator=g), 5, 1, 2) * 8
        xy = torch.rand(n, 2, generator=g) * 60 - 5
        boxes = torch.cat([xy, xy + torch.rand(n, 2, generator=g) * 50 + 3], 1)
        boxes[0] = torch.tensor([10.0, 10.0, 10.0, 30.0])             # zero-width box: the inf -> 0 branch
        labels = torch.randint(0, nc, (n,), generator=g)
        b_in = boxes.clone()
        fake = types.SimpleNamespace(class_agnostic=False)
        masks = fcn.FCNMaskHead._predict_by_feat_single(fake, logits.clone(), b_in, labels, meta,
                                                       mf.CD(dict(mask_thr_binary=0.5)), rescale=rescale)
        # :390-394 `threshold < 0`: the pasted probabilities as (p * 255) -> uint8
        soft = fcn.FCNMaskHead._predict_by_feat_single(fake, logits.clone(), boxes.clone(), labels, meta,
                                                      mf.CD(dict(mask_thr_binary=-1)), rescale=rescale)
        cases.append(dict(logits=logits, boxes=boxes, labels=labels, meta=meta, rescale=rescale, masks=masks,
                          masks_soft=soft, boxes_out=b_in))
    out['predict_single'] = cases
    g = torch.Generator().manual_seed(9)
    probs = torch.rand(4, 1, 28, 28, generator=g)
    bx = torch.tensor([[3.2, 4.1, 40.7, 33.3], [-6.0, 2.0, 20.0, 70.0], [10.0, 10.0, 12.0, 11.0], [0.0, 0.0, 64.0, 48.0]])
    pasted, _ = fcn._do_paste_mask(probs, bx, 48, 64, skip_empty=False)
    out['paste'] = dict(probs=probs, boxes=bx, img_hw=(48, 64), out=pasted.clone())

    # ------------------------------------------------------------------ the STANDARD Mask2FormerHead of SAMSegMask2Former
    # on the reference's own panoptic_head config (configs/rsprompter/samseg-mask2former-nwpu.py over its _base_)
    import rsprompter_amd as ra
    m2h, _, _ = mf.load_sources()
    qcfg = ra.Config.fromfile('/root/reference/configs/rsprompter/samseg-mask2former-nwpu.py').model
    ph = mf.CD({k: v for k, v in qcfg.panoptic_head.items() if k != 'type'})
    NQ, NC, Bq = 10, 3, 2
    ph.update(num_queries=NQ, num_things_classes=NC, train_cfg=None, test_cfg=None)
    ph['loss_cls'] = mf.CD(dict(ph['loss_cls'], class_weight=[1.0] * NC + [0.1]))
    head = mf.seeded(m2h.Mask2FormerHead(**ph), 32)
    sizes = (32, 16, 8, 4, 2)
    xs = [mf.rnd(310 + i, Bq, 256, s, s) for i, s in enumerate(sizes)]
    mask_features, memories = head.pixel_decoder(xs)
    cls_l, mask_l = head(xs, None)
    out['m2f_head'] = dict(keys=mf.keyshapes(head), seed=32, num_queries=NQ, num_classes=NC, batch=Bq,
                           xs=[(310 + i, (Bq, 256, s, s)) for i, s in enumerate(sizes)],
                           mask_features=mask_features[:, ::4, ::2, ::2].clone(), memories=[m[:, ::4].clone() for m in memories],
                           cls_pred_all=[c.clone() for c in cls_l], mask_pred_all=[m[:, :, ::2, ::2].clone() for m in mask_l],
                           mask_pred=mask_l[-1].clone())
    torch.save(out, OUT)
    print('wrote', OUT, os.path.getsize(OUT) // 1024, 'KiB')


if __name__ == '__main__':
    main()
