#!/usr/bin/env python
"""A/B timing of code-path switches on ONE model in ONE process (a bench.py run per arm costs a minute of model set-up each).

  python tools/ab_bench.py [--arch huge] [--batch 8] [--steps 4] [--rounds 2] "name:attr=value,attr=value" ...

Every arm sets attributes on every module of the model that has them (`t2i_fold`, `upscale_fused`) or module globals (`ops.X=...`), then times `steps` whole test_steps; the arms are interleaved `rounds` times so
that clock drift hits all of them.  Prints one line per arm: median ms per step, and the per-kernel HIP-event table of the
kernels whose time differs between the arms.  Synthetic weights / tiles exactly as bench.py builds them.
"""
import argparse
import json
import os
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def parse_arm(s):
    name, _, rest = s.partition(':')
    kv = {}
    for item in filter(None, rest.split(',')):
        k, _, v = item.partition('=')
        kv[k] = {'True': True, 'False': False}.get(v, int(v) if v.lstrip('-').isdigit() else v)
    return name, kv


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--arch', default='huge')
    ap.add_argument('--batch', type=int, default=8)
    ap.add_argument('--model', default='anchor')
    ap.add_argument('--steps', type=int, default=4)
    ap.add_argument('--rounds', type=int, default=2)
    ap.add_argument('--kernels', action='store_true', help='one profiled step per arm: per-kernel ms')
    ap.add_argument('arms', nargs='+')
    args = ap.parse_args()
    from rsprompter_amd import ops
    from rsprompter_amd.sam_decoder import SamMaskDecoderHIP
    from rsprompter_amd.sam_encoder import SamVisionEncoderHIP
    from rsprompter_amd.structures import DetDataSample
    from rsprompter_amd.synth import synth_images
    dev = torch.device('cuda', 0)
    model = bench.build_model(args.arch, 10 if args.model == 'anchor' else 1, dev, args.model, False)
    imgs = [im.to(dev) for im in synth_images(args.batch, seed=1234)]
    metas = bench.bench_metas(args.batch, args.model, False)
    targets = list(model.modules())          # an attribute is set on every module that has it

    def apply(kv):
        for k, v in kv.items():
            if k.startswith('ops.'):
                setattr(ops, k[4:], v)
                continue
            hit = 0
            for m in targets:
                if hasattr(m, k):
                    setattr(m, k, v)
                    hit += 1
            if not hit:
                raise SystemExit(f'no module has attribute {k}')

    def step():
        out = model.test_step(dict(inputs=imgs, data_samples=[DetDataSample(metainfo=dict(m)) for m in metas]))
        return out

    arms = [parse_arm(a) for a in args.arms]
    times = {n: [] for n, _ in arms}
    kern = {}
    for r in range(args.rounds):
        for name, kv in arms:
            apply(kv)
            step(); step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                step()
            torch.cuda.synchronize()
            times[name].append((time.perf_counter() - t0) / args.steps * 1e3)
            if args.kernels and r == 0:
                prof = ops.Profiler()
                ops.set_profiler(prof)
                step()
                torch.cuda.synchronize()
                ops.set_profiler(None)
                kern[name] = {k: round(v['ms'], 3) for k, v in prof.summary().items()}
    for name, _ in arms:
        print(f'{name:24s} median {statistics.median(times[name]):8.2f} ms/step   runs {[round(t, 2) for t in times[name]]}')
    if kern:
        keys = sorted({k for d in kern.values() for k in d}, key=lambda k: -max(d.get(k, 0) for d in kern.values()))
        for k in keys:
            vals = [kern[n].get(k, 0.0) for n, _ in arms]
            if max(vals) - min(vals) > 0.05:
                print(f'  {k[:70]:70s} ' + ' '.join(f'{v:8.3f}' for v in vals))
        print('  totals of the event tables: ' + ' '.join(f'{sum(kern[n].values()):8.2f}' for n, _ in arms))
    print(json.dumps({'times_ms': times}))


if __name__ == '__main__':
    main()
