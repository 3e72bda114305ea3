"""how many attention-mask decisions (models.py:381-392: sigmoid < 0.5) and how much mask-logit
error does the REFERENCE's own fp32 forward show against an fp64 forward of the same model, on the configs[4] fixture
(rsprompter_query, SAM ViT-H + LoRA, Nq = 100, WHU-shape; tests/test_gpu_baseline_configs.py::test_config4...)?

  python tools/parity_fp64_study.py [out.json]

The oracle (HF SAM modules + restated mmdet glue, pinned on the reference's classes) is run twice on the CPU: as the
reference runs it (fp32), and with every parameter / buffer / input in fp64.  The fp64 run is the "true" function of the
weights; what the fp32 run differs from it by is the reference's own round-off -- the floor any fp32-class implementation
(ours included) can be held to."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from oracle import glue
    from oracle.query import QueryOracle
    from rsprompter_amd.synth import synth_images, synth_metas, synth_state_dict
    out_path = sys.argv[1] if len(sys.argv) > 1 else 'profiles/r3_parity_fp64_study_config4.json'
    torch.manual_seed(0)
    arch = os.environ.get('STUDY_ARCH', 'huge')
    oracle = QueryOracle(arch, 1, 100, max_per_image=100, lora=dict(r=16, alpha=32))
    sd = synth_state_dict(oracle, seed=2)
    oracle.load_state_dict(sd)
    imgs = synth_images(1, seed=77)
    metas = synth_metas(1, ori_shape=(512, 512), scale_factor=(2.0, 2.0))
    x = glue.data_preprocess(imgs, [123.675, 116.28, 103.53], [58.395, 57.12, 57.375], True, 32)
    t = time.time()
    ref32, tr32 = oracle.predict(x, metas)
    t32 = time.time() - t
    print(f'fp32 forward {t32:.1f} s', flush=True)
    keep = lambda tr: dict(attn_masks=[m.clone() for m in tr['attn_masks']], mask_pred=tr['mask_pred'].clone(),
                           cls_pred=tr['cls_pred'].clone(), mpp=[m.clone() for m in tr['mask_pred_plus_all']],
                           emb=tr['image_embeddings'].clone())
    a = keep(tr32)
    del tr32
    oracle = oracle.double()
    t = time.time()
    torch.set_default_dtype(torch.float64)          # glue helpers that create tensors follow the default dtype
    ref64, tr64 = oracle.predict(x.double(), metas)
    torch.set_default_dtype(torch.float32)
    t64 = time.time() - t
    print(f'fp64 forward {t64:.1f} s', flush=True)
    b = keep(tr64)
    assert b['mask_pred'].dtype == torch.float64 and b['emb'].dtype == torch.float64
    flips, flipped_q = [], None
    nq = a['mask_pred'].shape[1]
    for m32, m64 in zip(a['attn_masks'], b['attn_masks']):
        d = (m32 != m64)
        flips.append(int(d.sum()) // 8)                                    # the mask is repeated over the 8 heads
        dq = d.view(1, -1, nq, d.shape[-1])[:, 0].any(-1)
        flipped_q = dq if flipped_q is None else (flipped_q | dq)
    per_q = (a['mask_pred'].double() - b['mask_pred']).abs().flatten(2).amax(2)[0]          # [Nq]
    aux_err = [float((p.double() - q).abs().max()) for p, q in zip(a['mpp'], b['mpp'])]
    n_dec = sum(int(m.numel()) // 8 for m in a['attn_masks'])
    res = dict(
        fixture='configs[4]: rsprompter_query SAM ViT-%s + LoRA(r16, alpha32), Nq=100, seed 2 weights, image seed 77' % arch,
        seconds=dict(fp32=round(t32, 1), fp64=round(t64, 1)), threads=torch.get_num_threads(),
        attention_mask_decisions=n_dec,
        fp32_vs_fp64=dict(
            attn_mask_bits_that_differ_per_layer=flips, total=sum(flips), queries_touched=int(flipped_q.sum()),
            aux_mask_logit_err_per_head_call=aux_err,
            sam_mask_logit_err_max=float(per_q.max()), sam_mask_logit_err_max_unflipped=float(per_q[~flipped_q[0]].max()),
            sam_mask_logit_err_median=float(per_q.median()),
            five_largest_per_query=[float(v) for v in per_q.topk(5).values],
            cls_logit_err=float((a['cls_pred'].double() - b['cls_pred']).abs().max()),
            image_embedding_err=float((a['emb'].double() - b['emb']).abs().max()),
            mask_logit_range=float(b['mask_pred'].abs().max())))
    print(json.dumps(res, indent=1))
    os.makedirs(os.path.dirname(out_path) or '.', exist_ok=True)
    with open(out_path, 'w') as f:
        json.dump(res, f, indent=1)


if __name__ == '__main__':
    main()
