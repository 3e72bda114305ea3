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
    torch.save(out, OUT)
    print('wrote', OUT, {k: (len(v) if isinstance(v, list) else list(v.keys())) for k, v in out.items()})


if __name__ == '__main__':
    main()
