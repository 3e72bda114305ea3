"""CPU oracle of the RSPrompter inference hot path -- TEST INFRASTRUCTURE ONLY.

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may
import this package.  Nothing under `rsprompter_amd/` imports it; the product
path fails loudly if librsp_hip.so is missing instead of falling back here.

What it is: a plain PyTorch (CPU, fp32 or fp64) restatement of the reference's
predict path.  SAM arithmetic comes straight from the reference's own pinned
third-party dependency, HuggingFace `transformers.models.sam` (reference pins
4.38.1, README.md:137; 5.15.0 is what is installed -- arithmetic of HF:803-831
and HF:461-543 is unchanged), forced to eager attention.
The mmdet / mmcv glue (RPN, anchors, box coder, RoIAlign, NMS, RoI heads,
post-processing) is restated from the cited reference lines because mmcv /
mmengine are not installable here.

Pinning status:
  * delta2bbox, AnchorGenerator, SinePositionalEncoding, window partition /
    rel-pos helpers of vit_sam.py, LN2d, RSFeatureAggregator, the mask
    post-process and the SAM positional embedding are checked against outputs
    of the REAL reference source files executed in the build container
    (tests/golden/make_golden.py -> tests/golden/*.pt) and against the two
    known-answer tests the reference inherits
    (test_delta_xywh_bbox_coder.py:9-24, test_anchor_generator.py:290-309).
  * query path: mask2bbox, MaskFormerFusionHead.instance_postprocess and RSMaskFormerFusionHead.predict are checked
    the same way (tests/golden/make_golden_query.py, order-free: the reference's topk is sorted=False).
  * detection glue of the anchor path -- RPNHead._predict_by_feat_single/_bbox_post_process, multiclass_nms,
    BBoxHead._predict_by_feat_single, SingleRoIExtractor.map_roi_levels -- is checked against the real reference
    sources run AROUND an injected batched_nms (tests/golden/make_golden_heads.py): everything that decides which
    indices come out except the NMS primitive itself (exact score ties compared order-free: the reference's sort is
    not stable).
  * COCO RLE (oracle/rle.py, pycocotools restated): pinned on the compressed RLE strings of the reference's
    tests/data/vis_sample.json (tests/golden/coco_rle_strings.json).
  * module FORWARDS the reference wires out of mmcv bricks -- RSSimpleFPN, PseudoFeatureAggregator (+ RSFPN),
    RSPrompterAnchorMaskHead.forward, RSMask2FormerHead.{_forward_head, forward} (incl. MSDeformAttnPixelDecoder.forward,
    Mask2FormerTransformerEncoder / Decoder layers, MlvlPointGenerator, SinePositionalEncoding) and ViTSAM.forward --
    are checked against the REAL classes executed in the build container on torch stand-ins for the mmcv leaves
    (tests/golden/make_golden_forwards.py + mmcv_standins.py -> tests/test_oracle_forwards.py), weights and inputs
    drawn from (seed, key, shape) on both sides, so the `state_dict` key layout (names, shapes) is pinned too.
    The standard Mask2FormerHead of SAMSegMask2Former (oracle/samseg.py) is pinned the same way on the real
    mmdet/models/dense_heads/mask2former_head.py (tests/golden/make_golden_samseg.py), FCNMaskHead + mask paste too.
    That run established that the LN2d layers of RSSimpleFPN carry eps = 1e-5 (mmcv `build_norm_layer` default), not
    LN2d's own 1e-6.
  * test pipeline front end (Resize keep_ratio + Pad, oracle/pipeline.py): restated from mmcv 2.1 / OpenCV documented
    behaviour, cross-checked against torch's independent bilinear; cv2 itself is not available: UNPINNED.
  * SAM encoder / mask decoder: HF modules themselves (the reference's dependency).
  * mmcv LEAF ops -- RoIAlign, nms / batched_nms, MultiScaleDeformableAttention, MultiheadAttention, FFN -- and peft's
    LoRA wrapping: source not in /root/reference -> restated from documented semantics and checked
    by hand-derivable known-answer vectors (tests/test_oracle_golden.py::test_mmcv_leaf_known_answers):
    PARITY UNPINNED at that boundary only.
"""
