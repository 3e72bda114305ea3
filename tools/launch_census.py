"""Round 6: which torch-native device launches does one step of the bench model issue, and from where (VERDICT r5 item 8:
~290 per ViT-H step, 36 % of all launches)?  One profiled step after warm-up; kernels grouped by name and by the innermost
rsprompter_amd source line on the Python stack.

    python tools/launch_census.py [--arch huge] [--model anchor] [--batch 8]"""
import argparse
import collections
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
    ap.add_argument('--lora', action='store_true')
    a = ap.parse_args()
    import bench
    from rsprompter_amd.structures import DetDataSample
    from rsprompter_amd.synth import synth_images
    dev = torch.device('cuda:0')
    model = bench.build_model(a.arch, 10 if a.model == 'anchor' else 1, dev, a.model, a.lora)
    imgs = [im.to(dev) for im in synth_images(a.batch, seed=1234)]
    metas = bench.bench_metas(a.batch, a.model, a.lora)

    def step():
        return model.test_step(dict(inputs=imgs, data_samples=[DetDataSample(metainfo=dict(m)) for m in metas]))
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        step()
        torch.cuda.synchronize()
    by_name, by_site, total, ours = collections.Counter(), collections.Counter(), 0, 0
    for ev in prof.events():
        if ev.device_type != torch.autograd.DeviceType.CUDA and not getattr(ev, 'kernels', None):
            continue
    # aten ops that launched something: count their device kernels, attribute to the innermost package frame
    for ev in prof.events():
        ks = getattr(ev, 'kernels', None) or []
        if not ks or not ev.name.startswith('aten::'):
            continue
        # only leaf aten ops (a parent's kernels are its children's)
        if any(getattr(c, 'kernels', None) for c in (ev.cpu_children or [])):
            continue
        site = next((f for f in (ev.stack or []) if 'rsprompter_amd' in f and 'ops.py' not in f), None) or \
            next((f for f in (ev.stack or []) if 'rsprompter_amd' in f), '?')
        site = site.split('rsprompter_amd/')[-1] if 'rsprompter_amd/' in site else site
        by_name[ev.name] += len(ks)
        by_site[(site.strip(), ev.name)] += len(ks)
        total += len(ks)
    n_all = sum(1 for ev in prof.events() if ev.device_type == torch.autograd.DeviceType.CUDA)
    print(f'{a.model} ViT-{a.arch} batch {a.batch}: {total} torch-native device launches in one step ({n_all} device events in all)')
    for n, c in by_name.most_common(25):
        print(f'  {c:5d}  {n}')
    print('by source line:')
    for (site, n), c in by_site.most_common(60):
        print(f'  {c:5d}  {n:28s} {site}')


if __name__ == '__main__':
    main()
