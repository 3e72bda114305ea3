"""Golden vectors for bench.py's parity canary: the CPU oracle's answer for tile 0 of the bench's own seeded fixture
(weight seed 0, images synth_images(B, seed=1234), rank 0), one file per bench configuration:

  python tests/golden/make_golden_bench.py anchor huge          -> bench_canary_anchor_huge.pt   (configs[3] slice)
  python tests/golden/make_golden_bench.py anchor base          -> bench_canary_anchor_base.pt   (configs[1])
  python tests/golden/make_golden_bench.py query large          -> bench_canary_query_large.pt   (configs[2])
  python tests/golden/make_golden_bench.py query huge --lora    -> bench_canary_query_huge_lora.pt (configs[4] slice,
                                                                   WHU-shape metas: ori_shape 512, scale_factor 2)

Stored (0.2-0.4 MB each).  anchor: the detections of the tile (boxes, scores, labels, kept-candidate indices), a strided sample of each
detection's 256x256 low-resolution SAM mask logits (every 16th row / column) and of the image embedding (every 8th
position).  query: class logits of all Nq queries, the same strided sample of every query's SAM mask logits, the selected
query indices, the embedding sample.  bench.py compares the HIP path's tile 0 with these numbers OUTSIDE its timed region
and prints the errors next to the throughput: a bench that runs on NaN rows or on a broken kernel says so itself."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

MEAN, STD = [123.675, 116.28, 103.53], [58.395, 57.12, 57.375]


def bench_metas(n, kind, lora):
    """the metas bench.py uses for this configuration (bench.py imports this function's twin: bench_metas there)"""
    from rsprompter_amd.synth import synth_metas
    if kind == 'query' and lora:
        return synth_metas(n, ori_shape=(512, 512), scale_factor=(2.0, 2.0))
    return synth_metas(n)


def canary_name(kind, arch, lora):
    return f'bench_canary_{kind}_{arch}' + ('_lora' if lora else '') + '.pt'


def main(kind='anchor', arch='huge', lora=False):
    from oracle import glue
    from rsprompter_amd.synth import synth_images, synth_state_dict
    t = time.time()
    imgs = synth_images(8, seed=1234)[:1]                 # bench.py: synth_images(B, seed=1234 + 1000 * rank), tile 0
    metas = bench_metas(1, kind, lora)
    x = glue.data_preprocess(imgs, MEAN, STD, True, 32)
    if kind == 'anchor':
        from oracle.anchor import AnchorOracle
        o = AnchorOracle(arch, 10)
        o.load_state_dict(synth_state_dict(o, seed=0))
        ref, tr = o.predict(x, metas)
        r = ref[0]
        low = tr['low_res_masks']                             # [k, 1, 256, 256]
        out = dict(kind=kind, arch=arch, image_seed=1234, weight_seed=0,
                   bboxes=r['bboxes'].float(), scores=r['scores'].float(), labels=r['labels'].long(),
                   cand=tr['dets'][0]['cand'].long(),            # kept candidates: index into the tile's (proposal, class) score matrix
                   low_res_sample=low[:, 0, ::16, ::16].contiguous().float(),
                   low_res_absmax=float(low.abs().max()))
        n = r['labels'].shape[0]
    else:
        from oracle.query import QueryOracle
        o = QueryOracle(arch, 1, 100, max_per_image=100, lora=dict(r=16, alpha=32) if lora else None)
        o.load_state_dict(synth_state_dict(o, seed=0))
        ref, tr = o.predict(x, metas)
        low = tr['mask_pred']                                 # [1, Nq, 256, 256]
        out = dict(kind=kind, arch=arch, lora=bool(lora), image_seed=1234, weight_seed=0,
                   cls_pred=tr['cls_pred'][0].float(), low_res_sample=low[0, :, ::16, ::16].contiguous().float(),
                   low_res_absmax=float(low.abs().max()), query_indices=ref[0]['query_indices'].long(),
                   scores=ref[0]['scores'].float())
        n = low.shape[1]
    out.update(embedding_sample=tr['image_embeddings'][0, :, ::8, ::8].contiguous().float(),
               embedding_absmax=float(tr['image_embeddings'].abs().max()))
    path = os.path.join(ROOT, 'tests', 'golden', canary_name(kind, arch, lora))
    torch.save(out, path)
    print(f'{path}: {n} prompt sets, logits range {out["low_res_absmax"]:.2f}, '
          f'{os.path.getsize(path) / 1e3:.0f} kB, {time.time() - t:.0f} s')


if __name__ == '__main__':
    pos = [a for a in sys.argv[1:] if not a.startswith('--')]
    if len(pos) == 1 and pos[0] in ('base', 'large', 'huge'):     # round-2 call form: arch only
        pos = ['anchor', pos[0]]
    main(pos[0] if pos else 'anchor', pos[1] if len(pos) > 1 else 'huge', '--lora' in sys.argv)
