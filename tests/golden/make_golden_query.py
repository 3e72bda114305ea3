"""Golden vectors for the query path's pure-torch reference functions, produced by executing the REAL reference
sources with the stub modules of make_golden.py:
  mask2bbox                        mmdet/structures/mask/utils.py:56-77
  MaskFormerFusionHead.instance_postprocess   mmdet/models/seg_heads/panoptic_fusion_heads/maskformer_fusion_head.py:126-182
Run in the build container:  python tests/golden/make_golden_query.py   ->  tests/golden/reference_vectors_query.pt
(`topk(sorted=False)` leaves the ORDER of the kept entries unspecified; the replay test compares order-free.)"""
import os
import sys
import types

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402

OUT = os.path.join(HERE, 'reference_vectors_query.pt')


def main():
    mg._install_stubs()
    for n in ('pycocotools', 'pycocotools.mask'):
        m = types.ModuleType(n)
        m.__path__ = []
        sys.modules[n] = m
    sys.modules['pycocotools'].mask = sys.modules['pycocotools.mask']
    utils = mg._load('mmdet/structures/mask/utils.py', '_ref_mask_utils')
    out = {}
    g = torch.Generator().manual_seed(3)
    masks = torch.rand(9, 40, 56, generator=g) > 0.97
    masks[2] = False                                   # empty mask -> zero box (utils.py:70-75)
    masks[3] = False; masks[3, 39, 55] = True          # single pixel in the corner
    masks[4] = True
    out['mask2bbox'] = dict(masks=masks, out=utils.mask2bbox(masks))

    sys.modules['mmdet.structures.mask'].mask2bbox = utils.mask2bbox
    fh = mg._load('mmdet/models/seg_heads/panoptic_fusion_heads/maskformer_fusion_head.py', '_ref_fusion')
    cases = []
    for (nq, nc, k, hw, seed) in ((30, 1, 20, (32, 48), 0), (50, 10, 100, (24, 24), 1), (12, 3, 100, (16, 16), 2)):
        g = torch.Generator().manual_seed(10 + seed)
        head = fh.MaskFormerFusionHead()
        head.num_things_classes, head.num_stuff_classes, head.num_classes = nc, 0, nc
        head.test_cfg = dict(max_per_image=min(k, nq * nc))
        mask_cls = torch.randn(nq, nc + 1, generator=g) * 2
        mask_pred = torch.randn(nq, *hw, generator=g) * 3
        mask_pred[1] = -1.0                             # a query with an empty mask
        r = head.instance_postprocess(mask_cls, mask_pred)
        cases.append(dict(num_classes=nc, max_per_image=head.test_cfg['max_per_image'], mask_cls=mask_cls,
                          mask_pred=mask_pred, bboxes=r.bboxes, labels=r.labels, scores=r.scores, masks=r.masks))
    out['instance_postprocess'] = cases

    # RSMaskFormerFusionHead.predict (mmdet/rsprompter/models.py:663-715) on top of the real fusion head
    import transformers  # noqa: F401  (real; models.py imports it)
    sys.modules['mmdet.models'].MaskFormerFusionHead = fh.MaskFormerFusionHead
    models = mg._load('mmdet/rsprompter/models.py', '_ref_models')
    preds = []
    for (meta, nq, nc, seed) in ((dict(img_shape=(64, 64), ori_shape=(64, 64), scale_factor=(1.0, 1.0)), 20, 1, 0),
                                 (dict(img_shape=(64, 64), ori_shape=(32, 32), scale_factor=(2.0, 2.0)), 20, 2, 1),
                                 (dict(img_shape=(64, 64), ori_shape=(40, 27), scale_factor=(1.5, 1.5)), 16, 3, 2)):
        g = torch.Generator().manual_seed(40 + seed)
        head = models.RSMaskFormerFusionHead()
        head.num_things_classes, head.num_stuff_classes, head.num_classes = nc, 0, nc
        head.test_cfg = dict(max_per_image=10, panoptic_on=False, semantic_on=False, instance_on=True)
        mask_cls = torch.randn(1, nq, nc + 1, generator=g) * 2
        mask_pred = torch.randn(1, nq, 64, 64, generator=g) * 3          # already at batch_input_shape (models.py:652-656)
        r = head.predict(mask_cls, mask_pred, [types.SimpleNamespace(metainfo=meta)], rescale=True)[0]['ins_results']
        preds.append(dict(meta=meta, num_classes=nc, max_per_image=10, mask_cls=mask_cls, mask_pred=mask_pred,
                          bboxes=r.bboxes, labels=r.labels, scores=r.scores, masks=r.masks))
    out['fusion_predict'] = preds
    torch.save(out, OUT)
    print('wrote', OUT, {k: (len(v) if isinstance(v, list) else list(v.keys())) for k, v in out.items()})


if __name__ == '__main__':
    main()
