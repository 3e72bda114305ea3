#!/usr/bin/env python
"""bench.py -- RSPrompter inference hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py ...)

A "step" is one pass of the whole hot path (DetDataPreprocessor -> SAM ViT encoder -> RSFPN -> RPN ->
RoI prompter -> SAM mask decoder -> mask post-process -> result all-gather when N > 1) over one batch of
synthetic 1024x1024 tiles already resident in HBM.  The workload is BASELINE.json configs[1]:
rsprompter_anchor, SAM ViT-B, batch 8 x 1024 x 1024 per GPU (weak scaling: every rank runs its own batch).
Weights are seeded synthetic tensors of the reference architecture (there is no checkpoint / dataset).
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time
import warnings

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F16_MFMA_TFLOPS = 2500.0   # dense fp16/bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
ATTN_GEMM_GFLOP_PER_IMAGE = {'base': 229.8, 'large': 353.6, 'huge': 481.3}   # SURVEY.md §8d / BASELINE.md §3


def build_model(arch, num_classes, device, kind='anchor'):
    import rsprompter_amd as ra
    from rsprompter_amd.default_configs import rsprompter_anchor, rsprompter_query
    from rsprompter_amd.synth import synth_state_dict
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        cfg = rsprompter_anchor(arch, num_classes) if kind == 'anchor' else rsprompter_query(arch, num_classes)
        model = ra.build_model(cfg)
    model.load_state_dict(synth_state_dict(model, seed=0), strict=True)
    return model.to(device)


def cpu_baseline(arch, num_classes):
    """The reference path restated on the CPU (oracle/) timed on this box's host cores; bounded sample."""
    from oracle import glue
    from oracle.anchor import AnchorOracle
    from rsprompter_amd.synth import synth_images, synth_metas, synth_state_dict
    o = AnchorOracle(arch, num_classes)
    o.load_state_dict(synth_state_dict(o, seed=0))
    n = 1
    x = glue.data_preprocess(synth_images(n), [123.675, 116.28, 103.53], [58.395, 57.12, 57.375], True, 32)
    t = time.perf_counter()
    o.predict(x, synth_metas(n))
    dt = time.perf_counter() - t
    return dict(value=n / dt, unit='images/s', cores=torch.get_num_threads(), kind='port',
                sample=f'{n} x 1024x1024 tile through the full CPU oracle (fp32 PyTorch, HF SAM eager attention), '
                       f'single timed pass of {dt:.1f} s, no warm-up')


def _pmc_traffic():
    """HBM bytes per launch of the dominant GEMM from the committed rocprofv3 PMC passes (profiles/r1_pmc/,
    FETCH_SIZE x2 + WRITE_SIZE as MI355X_MICROARCH.md prescribes); the counters cannot be collected inside this
    process, so the number is the one measured with tools/pmc_gemm.sh on the kernel's most frequent shape."""
    f = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles', 'r1_pmc', 'gemm_lin1_traffic.json')
    try:
        with open(f) as fh:
            t = json.load(fh)
        return {'bytes_per_launch': t['traffic_bytes_per_launch'], 'algorithmic_bytes_per_launch': t['algorithmic_bytes_per_launch'],
                'shape': t['shape'], 'source': 'profiles/r1_pmc/gemm_lin1_traffic.json'}
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--arch', default='base')
    ap.add_argument('--batch', type=int, default=8, help='images per GPU per step')
    ap.add_argument('--model', default='anchor', choices=['anchor', 'query'],
                    help="prompter variant; the headline line is 'anchor' (BASELINE.json configs[1])")
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--shapes', action='store_true', help='break the kernel table down by GEMM/attention shape')
    args = ap.parse_args()

    from rsprompter_amd import dist as rdist
    from rsprompter_amd import ops
    from rsprompter_amd.structures import DetDataSample
    from rsprompter_amd.synth import synth_images, synth_metas
    import torch.distributed as tdist

    rank, local, world = rdist.init_from_env()
    if world != args.gpus:
        if rank == 0:
            print(f'warning: --gpus {args.gpus} but WORLD_SIZE={world}', file=sys.stderr)
    if not torch.cuda.is_available():
        raise RuntimeError('bench.py needs an MI355X (no CPU fallback)')
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    num_classes = 10 if args.model == 'anchor' else 1
    model = build_model(args.arch, num_classes, dev, args.model)
    B = args.batch
    imgs = [im.to(dev) for im in synth_images(B, seed=1234 + 1000 * rank)]
    metas = synth_metas(B)

    def step():
        samples = [DetDataSample(metainfo=dict(m)) for m in metas]
        out = model.test_step(dict(inputs=imgs, data_samples=samples))
        res = [o.pred_instances for o in out]
        if world > 1:
            rdist.all_gather_results(res)
        return res

    def sync():
        if world > 1:
            tdist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        tdist.all_reduce(t, op=tdist.ReduceOp.MAX)
        elapsed = float(t.item())
    n_dets = sum(len(r.bboxes) for r in step())

    result = None
    # ---- roofline leg: one extra instrumented step, per-kernel HIP events on the launch stream.  Every rank runs
    # the step (it contains the result all-gather); only rank 0 records and reports ----
    prof = ops.Profiler() if rank == 0 else None
    if prof is not None:
        prof.shapes = args.shapes
        ops.set_profiler(prof)
    step()
    ops.set_profiler(None)
    if rank == 0:
        agg = prof.summary()
        kernels = {k: dict(ms=round(v['ms'], 3), calls=v['calls'],
                           tflops=round(v['flops'] / v['ms'] / 1e9, 1) if v['ms'] > 0 and v['flops'] else None,
                           gbps=round(v['bytes'] / v['ms'] / 1e6, 1) if v['ms'] > 0 and v['bytes'] else None)
                   for k, v in sorted(agg.items(), key=lambda kv: -kv[1]['ms'])}
        # the dominant kernel is the plane-path GEMM: one kernel template whose tile instantiations (128x128,
        # 256x128, 256x256, ConvTranspose epilogues) the profiler labels separately; the roofline line covers all
        # of its launches in the step (the per-instantiation rates are in `kernels` and in the rocprofv3 stats)
        fam = {}
        for k, v in agg.items():
            if v['flops'] > 0:
                f = k.split('<')[0].split(' ')[0]
                d0 = fam.setdefault(f, dict(ms=0.0, flops=0.0, calls=0))
                d0['ms'] += v['ms']; d0['flops'] += v['flops']; d0['calls'] += v['calls']
        dom_name, dom = max(fam.items(), key=lambda kv: kv[1]['ms'])
        achieved = dom['flops'] / dom['ms'] / 1e9
        attn = [v for k, v in agg.items() if k.startswith('attn_kernel<vit') or k.startswith('attn_global_kernel')]
        attn_ms = sum(v['ms'] for v in attn)
        attn_tf = sum(v['flops'] for v in attn) / attn_ms / 1e9 if attn_ms else None
        value = world * B * args.steps / elapsed
        result = {
            'metric': 'images/sec (1024x1024 synthetic tiles, rsprompter_%s SAM-ViT-%s, full predict path)' % (args.model, args.arch[0].upper()),
            'value': round(value, 3), 'unit': 'images/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(elapsed / args.steps * 1e3, 3), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None,
            'dtype': 'f32 (fp32 in/out; GEMMs and attention as fp16x3 split-precision MFMA with fp32 accumulate)',
            'data': 'synthetic',
            'config': {'workload': f'rsprompter_{args.model} SAM-ViT-{args.arch}, batch {B}x1024x1024 per GPU, '
                                   f'{num_classes} classes, seeded synthetic weights' + (' (BASELINE.json configs[1])' if args.model == 'anchor' and args.arch == 'base' and B == 8 else ''),
                       'images_per_gpu_per_step': B, 'detections_per_step_rank0': n_dets,
                       'parallelism': f'dp{world} (images sharded by batch, result all-gather)' if world > 1 else 'single GPU'},
            'roofline': {'bound': 'mfma', 'kernel': dom_name, 'launches_per_step': dom['calls'],
                         'ms_per_step': round(dom['ms'], 3), 'achieved': round(achieved, 2),
                         'peak': PEAK_F16_MFMA_TFLOPS, 'unit': 'TFLOP/s', 'frac': round(achieved / PEAK_F16_MFMA_TFLOPS, 4),
                         'traffic': (_pmc_traffic() or {}).get('bytes_per_launch'), 'traffic_detail': _pmc_traffic(),
                         'note': 'achieved = algorithmic fp32 FLOPs (2MNK) of all launches of the kernel in one step / '
                                 'their summed HIP-event durations; the kernel issues 3 fp16 MFMA passes per algorithmic '
                                 'FLOP (fp16x3), so its ceiling is peak/3 = 833 TFLOP/s',
                         'frac_of_fp16x3_ceiling': round(achieved / (PEAK_F16_MFMA_TFLOPS / 3), 4)},
            'roofline_attention': {'bound': 'mfma', 'kernel': 'attn_kernel<vit>', 'achieved': None if attn_tf is None else round(attn_tf, 2),
                                   'peak': PEAK_F16_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                                   'frac': None if attn_tf is None else round(attn_tf / PEAK_F16_MFMA_TFLOPS, 4),
                                   'attention_gemm_gflop_per_image': ATTN_GEMM_GFLOP_PER_IMAGE[args.arch]},
            'kernels': kernels,
        }
        if world == 1 and not args.no_cpu_baseline and args.model == 'anchor':
            try:
                result['cpu_baseline'] = cpu_baseline(args.arch, num_classes)
            except Exception as e:  # the baseline is context, never a reason to lose the measurement
                result['cpu_baseline'] = {'value': None, 'unit': 'images/s', 'cores': torch.get_num_threads(),
                                          'kind': 'port', 'sample': f'failed: {e!r}'}
        print(json.dumps(result), flush=True)
    if world > 1:
        tdist.barrier()
        tdist.destroy_process_group()


if __name__ == '__main__':
    main()
