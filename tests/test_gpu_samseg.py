"""-m gpu: the SAMSeg sibling model `SAMSegMaskRCNN` (mmdet/rsprompter/models.py:1219-1244) on the HIP
kernels against its CPU oracle (oracle/samseg.py, mask branch pinned on the real fcn_mask_head.py)."""
import os
import sys
import warnings

import pytest
import torch

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _match import match_detections  # noqa: E402

MEAN = [123.675, 116.28, 103.53]
STD = [58.395, 57.12, 57.375]


def test_paste_masks_kernel_matches_reference_vectors(dev):
    """rsp_paste_masks against the outputs of the real FCNMaskHead._predict_by_feat_single (golden)."""
    from rsprompter_amd import ops
    g = torch.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'reference_vectors_samseg.pt'),
                   weights_only=False)
    for c in g['predict_single']:
        meta = c['meta']
        h, w = c['masks'].shape[-2:]
        got = ops.paste_masks(c['logits'].permute(0, 2, 3, 1).contiguous().to(dev), c['labels'].to(dev),
                              c['boxes_out'].to(dev), (h, w), 0.5)
        mism = float((got.cpu() != c['masks']).float().mean())
        print(f'paste {meta}: mismatch {mism:.2e}')
        assert mism < 1e-4          # pixels whose pasted probability sits within fp32 noise of the threshold


def test_samseg_maskrcnn_end_to_end(dev):
    import rsprompter_amd as ra
    from oracle import glue
    from oracle.samseg import SAMSegMaskRCNNOracle
    from rsprompter_amd.default_configs import samseg_maskrcnn
    from rsprompter_amd.structures import DetDataSample
    from rsprompter_amd.synth import synth_images, synth_metas, synth_state_dict
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        model = ra.build_model(samseg_maskrcnn('base', 10))
    oracle = SAMSegMaskRCNNOracle('base', 10)
    sd = synth_state_dict(oracle, seed=0)
    oracle.load_state_dict(sd)
    res = model.load_state_dict(sd, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    model = model.to(dev)
    imgs = synth_images(2)
    metas = synth_metas(2, ori_shape=(512, 512), scale_factor=(2.0, 2.0))
    x = glue.data_preprocess(imgs, MEAN, STD, True, 32)
    ref, tr = oracle.predict(x, metas)
    # mask logits on the oracle's RoI features (stage test), then the free-running pipeline
    lg = model.roi_head.mask_head(tr['mask_feats'].to(dev).contiguous(memory_format=torch.channels_last))
    e = float((lg.cpu() - tr['mask_logits']).abs().max())
    print('FCN mask head logits err %.2e (range %.1f)' % (e, float(tr['mask_logits'].abs().max())))
    assert e < 1e-3
    out = model.test_step(dict(inputs=[i.to(dev) for i in imgs], data_samples=[DetDataSample(metainfo=dict(m)) for m in metas]))
    for b in range(2):
        pi, r = out[b].pred_instances, ref[b]
        assert tuple(pi.masks.shape[1:]) == (512, 512) and pi.labels.shape[0] == r['labels'].shape[0]
        pairs = match_detections(pi.bboxes, pi.scores, pi.labels, r['bboxes'], r['scores'], r['labels'])
        ii = torch.tensor([i for i, _ in pairs]); jj = torch.tensor([j for _, j in pairs])
        mism = float((pi.masks.cpu()[ii] != r['masks'][jj]).float().mean())
        print(f'SAMSegMaskRCNN img {b}: {pi.labels.shape[0]} dets, {len(pairs)} matched, mask mismatch {mism:.2e}')
        assert mism < 1e-3
