"""Round 6: does ANY kernel of the product path give a different answer from run to run?  (The rare wrong answer of
sam_upscale2_kernel was found because one test happened to look at its output on an unlucky box; every other kernel of the
hot path is deterministic by construction -- no atomics on floating point, fixed reduction orders -- so repeated steps on the
same inputs must agree BIT FOR BIT.)

    python tools/determinism_soak.py [--arch huge] [--model anchor] [--batch 8] [--steps 100] [--lora]

Runs the bench's model on the bench's inputs `steps` times, torch.cuda.empty_cache() before every 4th step (fresh device
memory: the condition that raised the failure rate of the known case), and compares with the first step: the image
embedding, the SAM low-resolution mask logits, boxes / scores / labels / masks of every image.  Test infrastructure."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--arch', default='huge')
    ap.add_argument('--model', default='anchor')
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--lora', action='store_true')
    a = ap.parse_args()
    import bench
    import rsprompter_amd.debug as dbg
    from rsprompter_amd.structures import DetDataSample
    from rsprompter_amd.synth import synth_images
    dbg.KEEP_TRACES = True
    dev = torch.device('cuda:0')
    model = bench.build_model(a.arch, 10 if a.model == 'anchor' else 1, dev, a.model, a.lora)
    imgs = [im.to(dev) for im in synth_images(a.batch, seed=1234)]
    metas = bench.bench_metas(a.batch, a.model, a.lora)

    def step():
        out = model.test_step(dict(inputs=imgs, data_samples=[DetDataSample(metainfo=dict(m)) for m in metas]))
        t = {'embedding': model._last_embeddings}
        if a.model == 'anchor':
            t['low_res_masks'] = model.roi_head._last_mask_trace['mask_preds']
        else:
            cls, lazy = model._last_head_out
            t['class_logits'], t['low_res_masks'] = cls, lazy.low_res
        for b, o in enumerate(out):
            pi = o.pred_instances
            t[f'img{b}.bboxes'], t[f'img{b}.scores'], t[f'img{b}.labels'], t[f'img{b}.masks'] = pi.bboxes, pi.scores, pi.labels, pi.masks
        return {k: v.detach().clone() for k, v in t.items()}

    first = step()
    bad = 0
    for it in range(1, a.steps):
        if it % 4 == 3:
            torch.cuda.empty_cache()
        cur = step()
        torch.cuda.synchronize()
        moved = [k for k in first if cur[k].shape != first[k].shape or not torch.equal(cur[k], first[k])]
        if moved:
            bad += 1
            for k in moved[:4]:
                if cur[k].shape == first[k].shape:
                    d = (cur[k] != first[k])
                    idx = d.nonzero()
                    print(f'step {it}: {k} {tuple(cur[k].shape)}: {int(d.sum())} values moved, first at {idx[0].tolist()}, last at {idx[-1].tolist()}, '
                          f'max |diff| {float((cur[k].float() - first[k].float()).abs().max()):.3e}', flush=True)
                else:
                    print(f'step {it}: {k} shape {tuple(cur[k].shape)} vs {tuple(first[k].shape)}', flush=True)
    print(f'{a.model} ViT-{a.arch}{" + LoRA" if a.lora else ""} batch {a.batch}: {bad} of {a.steps - 1} repeated steps differ from the first '
          f'in any of {len(first)} tensors (bitwise)', flush=True)


if __name__ == '__main__':
    main()
