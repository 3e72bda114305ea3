"""The anchor-path companion of tools/parity_fp64_study.py: the REFERENCE's own fp32 forward against an fp64 forward of
the same model on the bench fixture (rsprompter_anchor, SAM ViT-H, 10 classes, weight seed 0, tile 0 of
synth_images(seed=1234) -- the fixture of bench.py's parity canary, tests/golden/bench_canary_anchor_huge.pt).

  python tools/parity_fp64_study_anchor.py [out.json]

The anchor path has no thresholded feedback (the query path's attention masks), so its fp32 forward is well defined:
what this measures is the size of fp32 round-off itself at every stage boundary -- the scale against which the HIP
path's errors (image embedding 1.0e-5, matched mask logits 4.7e-5 in bench.py's canary) are to be read -- and how many
discrete decisions (proposal / detection identities) the two precisions disagree on."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))


def main():
    from _match import match_detections
    from oracle import glue
    from oracle.anchor import AnchorOracle
    from rsprompter_amd.synth import synth_images, synth_metas, synth_state_dict
    out_path = sys.argv[1] if len(sys.argv) > 1 else 'profiles/r3_parity_fp64_study_anchor_vith.json'
    arch = os.environ.get('STUDY_ARCH', 'huge')
    torch.manual_seed(0)
    o = AnchorOracle(arch, 10)
    o.load_state_dict(synth_state_dict(o, seed=0))
    imgs = synth_images(8, seed=1234)[:1]
    metas = synth_metas(1)
    x = glue.data_preprocess(imgs, [123.675, 116.28, 103.53], [58.395, 57.12, 57.375], True, 32)
    t = time.time()
    r32, t32 = o.predict(x, metas)
    s32 = time.time() - t
    print(f'fp32 forward {s32:.1f} s', flush=True)
    keep = lambda tr: dict(emb=tr['image_embeddings'].clone(), fpn=[f.clone() for f in tr['fpn']],
                           rpn_cls=[c.clone() for c in tr['cls']], cls_score=tr['cls_score'].clone(),
                           props=tr['proposals'][0]['bboxes'].clone(), low=tr['low_res_masks'].clone())
    a, ra = keep(t32), r32[0]
    del t32
    o = o.double()
    # mmcv's RoIAlign / nms are C restatements in fp32 (oracle/mmcv_ops.c): in the fp64 run they read fp64 inputs rounded to
    # fp32 and their results are widened again -- one fp32 rounding of the sampled features (6e-8 relative), not a forward
    from oracle import cops
    _ra, _nms = cops.roi_align, cops.nms
    cops.roi_align = lambda feat, rois, *a_, **k_: _ra(feat, rois, *a_, **k_).to(feat.dtype)

    def nms_wide(boxes, scores, thr):
        _, keep = _nms(boxes, scores, thr)
        return torch.cat([boxes[keep], scores[keep, None]], 1), keep
    cops.nms = nms_wide
    torch.set_default_dtype(torch.float64)
    t = time.time()
    r64, t64 = o.predict(x.double(), metas)
    torch.set_default_dtype(torch.float32)
    s64 = time.time() - t
    print(f'fp64 forward {s64:.1f} s', flush=True)
    b, rb = keep(t64), r64[0]
    err = lambda p, q: float((p.double() - q.double()).abs().max())
    n_same_props = int((a['props'].double()[:, None, :] - b['props'][None]).abs().amax(-1).min(1).values.lt(1e-2).sum())
    out = dict(fixture='bench canary fixture: rsprompter_anchor SAM ViT-%s, 10 classes, weight seed 0, tile 0 of synth_images(seed=1234)' % arch,
               seconds=dict(fp32=round(s32, 1), fp64=round(s64, 1)), threads=torch.get_num_threads(),
               fp32_vs_fp64=dict(
                   image_embedding_err=err(a['emb'], b['emb']), image_embedding_range=float(b['emb'].abs().max()),
                   fpn_err_per_level=[err(p, q) for p, q in zip(a['fpn'], b['fpn'])],
                   rpn_objectness_logit_err_per_level=[err(p, q) for p, q in zip(a['rpn_cls'], b['rpn_cls'])],
                   proposals=int(a['props'].shape[0]), proposals_fp64=int(b['props'].shape[0]),
                   proposals_with_a_twin_within_1e_2_px=n_same_props,
                   detections=int(ra['labels'].shape[0]), detections_fp64=int(rb['labels'].shape[0])))
    if ra['labels'].shape[0] == rb['labels'].shape[0]:
        pairs = match_detections(ra['bboxes'], ra['scores'], ra['labels'], rb['bboxes'].float(), rb['scores'].float(), rb['labels'])
        ii = torch.tensor([i for i, _ in pairs]); jj = torch.tensor([j for _, j in pairs])
        out['fp32_vs_fp64'].update(
            detections_matched=len(pairs),
            detection_rank_changes=int((ii != jj).sum()),
            box_err_px=err(ra['bboxes'][ii], rb['bboxes'][jj]), score_err=err(ra['scores'][ii], rb['scores'][jj]),
            sam_mask_logit_err_max=err(a['low'][ii], b['low'][jj]), mask_logit_range=float(b['low'].abs().max()),
            final_mask_pixel_mismatch=float((ra['masks'][ii] != rb['masks'][jj]).float().mean()))
    with open(out_path, 'w') as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
