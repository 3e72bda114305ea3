"""Golden vector for bench.py's parity canary: the CPU oracle's answer for tile 0 of the bench's
own seeded fixture (rsprompter_anchor SAM ViT-H, 10 classes, weight seed 0, images synth_images(B, seed=1234), rank 0).

  python tests/golden/make_golden_bench.py            -> tests/golden/bench_canary_anchor_huge.pt  (~0.2 MB)

Stored: the detections of the tile (boxes, scores, labels), a strided sample of each detection's 256x256 low-resolution
SAM mask logits (every 16th row / column: 16 x 16 values per detection) and of the image embedding (every 8th position).
bench.py compares the HIP path's tile 0 with these numbers OUTSIDE its timed region and prints the errors next to the
throughput: a bench that runs on NaN rows or on a broken kernel says so itself."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main(arch='huge'):
    from oracle import glue
    from oracle.anchor import AnchorOracle
    from rsprompter_amd.synth import synth_images, synth_metas, synth_state_dict
    t = time.time()
    o = AnchorOracle(arch, 10)
    o.load_state_dict(synth_state_dict(o, seed=0))
    imgs = synth_images(8, seed=1234)[:1]                 # bench.py: synth_images(B, seed=1234 + 1000 * rank), tile 0
    metas = synth_metas(1)
    x = glue.data_preprocess(imgs, [123.675, 116.28, 103.53], [58.395, 57.12, 57.375], True, 32)
    ref, tr = o.predict(x, metas)
    r = ref[0]
    low = tr['low_res_masks']                             # [k, 1, 256, 256]
    out = dict(arch=arch, image_seed=1234, weight_seed=0,
               bboxes=r['bboxes'].float(), scores=r['scores'].float(), labels=r['labels'].long(),
               low_res_sample=low[:, 0, ::16, ::16].contiguous().float(),
               low_res_absmax=float(low.abs().max()),
               embedding_sample=tr['image_embeddings'][0, :, ::8, ::8].contiguous().float(),
               embedding_absmax=float(tr['image_embeddings'].abs().max()))
    path = os.path.join(ROOT, 'tests', 'golden', f'bench_canary_anchor_{arch}.pt')
    torch.save(out, path)
    print(f'{path}: {r["labels"].shape[0]} detections, logits range {out["low_res_absmax"]:.2f}, '
          f'{os.path.getsize(path) / 1e3:.0f} kB, {time.time() - t:.0f} s')


if __name__ == '__main__':
    main(sys.argv[1] if len(sys.argv) > 1 else 'huge')
