"""Golden vectors for the RSMask2FormerHead constructor branches no shipped config selects, produced by executing the REAL
`mmdet/rsprompter/models.py::RSMask2FormerHead` (and the real Mask2FormerHead.__init__ / MSDeformAttnPixelDecoder it
builds on) exactly as make_golden_forwards.py does for the shipped configuration:
  num_transformer_feat_level = 2 | 4 (= pixel_decoder num_levels)   mask2former_head.py:103-135; models.py:404-409,438,457
  enforce_decoder_input_project=True                               mask2former_head.py:118-128; models.py:405-406
  with_sincos=False                                                models.py:315-318, 346-347
  decoder_plus=False (one image: :365 expands by img_bs)           models.py:303-307, 361-385
  multimask_output=True (three masks folded into the query axis)   models.py:369-380; with decoder_plus=False the real class
                                                                   raises in its first decoder layer: the error is recorded
Run in the build container:  python tests/golden/make_golden_query_options.py -> tests/golden/reference_vectors_query_options.pt
Weights and inputs are pure functions of (seed, key, shape); only outputs are stored (strided)."""
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
import make_golden_forwards as mf  # noqa: E402

OUT = os.path.join(HERE, 'reference_vectors_query_options.pt')

# name -> (oracle QueryHead keywords, batch)
CASES = dict(
    levels2=(dict(levels=2), 2),
    levels4_proj=(dict(levels=4, input_proj=True), 2),
    no_sincos=(dict(with_sincos=False), 2),
    no_decoder_plus=(dict(decoder_plus=False), 1),
    multimask=(dict(multimask_output=True), 2),
)


@torch.no_grad()
def main():
    mf.install()
    _, _, models = mf.load_sources()
    import rsprompter_amd as ra
    qcfg = ra.Config.fromfile('/root/reference/configs/rsprompter/rsprompter_query-nwpu.py').model
    NQ, NC = 10, 2
    out = {}
    for n, (name, (kw, Bq)) in enumerate(CASES.items()):
        ph = mf.CD({k: v for k, v in qcfg.panoptic_head.items() if k != 'type'})
        ph.update(num_queries=NQ, num_things_classes=NC, train_cfg=None, test_cfg=None)
        ph['mask_decoder'] = mf.CD(dict(ph['mask_decoder'], init_cfg=None))
        ph['loss_cls'] = mf.CD(dict(ph['loss_cls'], class_weight=[1.0] * NC + [0.1]))
        levels = kw.get('levels', 3)
        ph['num_transformer_feat_level'] = levels
        ph['pixel_decoder']['encoder']['layer_cfg']['self_attn_cfg']['num_levels'] = levels
        ph['pixel_decoder']['num_outs'] = max(levels, 3)
        ph['enforce_decoder_input_project'] = kw.get('input_proj', False)
        ph['with_sincos'] = kw.get('with_sincos', True)
        ph['decoder_plus'] = kw.get('decoder_plus', True)
        ph['multimask_output'] = kw.get('multimask_output', False)
        head = mf.seeded(models.RSMask2FormerHead(**ph), 30 + n)
        xs_spec = [(200 + 10 * n + i, (Bq, 256, s, s)) for i, s in enumerate((64, 32, 16, 8, 4))]
        emb_spec, pe_spec = (260 + n, (Bq, 256, 16, 16)), (270 + n, (1, 256, 16, 16))
        xs = [mf.rnd(sd, *sh) for sd, sh in xs_spec]
        emb = mf.rnd(emb_spec[0], *emb_spec[1])
        pe = mf.rnd(pe_spec[0], *pe_spec[1]).repeat(Bq, 1, 1, 1)
        mask_features, memories = head.pixel_decoder(xs)
        cls_l, mask_l, mpp_l = head(xs, None, emb, pe)
        g = dict(keys=mf.keyshapes(head), seed=30 + n, num_queries=NQ, num_classes=NC, batch=Bq, head_kwargs=kw,
                 xs=xs_spec, emb=emb_spec, pe=pe_spec, n_memories=len(memories),
                 mask_features=mask_features[:, ::4, ::4, ::4].clone(), memories=[m[:, ::2].clone() for m in memories],
                 cls_pred_all=[c.clone() for c in cls_l], mask_pred=mask_l[-1][:, :, ::2, ::2].clone())
        if kw.get('decoder_plus', True):
            g['mask_pred_plus_all'] = [m[:, :, ::4, ::4].clone() for m in mpp_l]
        else:                                   # the SAM decoder ran in every stage: its masks are the attention-mask source
            g['mask_pred_all'] = [m[:, :, ::4, ::4].clone() for m in mask_l]
        out[name] = g
        print(name, len(g['keys']), 'state_dict keys,', len(memories), 'memories, mask_pred', tuple(mask_l[-1].shape))
    # multimask_output=True with decoder_plus=False: what the real class does (the folded [B, 3 Nq, h, w] masks become the
    # cross-attention mask of Nq queries)
    ph['decoder_plus'] = False
    head = mf.seeded(models.RSMask2FormerHead(**ph), 40)
    try:
        head([x[:1] for x in xs], None, emb[:1], pe[:1])
        raised = None
    except Exception as e:  # noqa: BLE001
        raised = dict(type=type(e).__name__, message=str(e)[:200])
    out['multimask_no_decoder_plus'] = dict(raised=raised)
    print('multimask_output=True, decoder_plus=False:', raised)
    torch.save(out, OUT)
    print('wrote', OUT, os.path.getsize(OUT) // 1024, 'KiB')


if __name__ == '__main__':
    main()
