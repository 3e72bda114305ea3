#!/usr/bin/env python
"""bench.py -- RSPrompter inference hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py ...)

A "step" is one pass of the whole hot path (DetDataPreprocessor -> SAM ViT encoder -> RSFPN -> RPN ->
RoI prompter -> SAM mask decoder -> mask post-process -> result gather to rank 0 when N > 1) over one batch of
synthetic 1024x1024 tiles already resident in HBM.  The default workload is the configuration BASELINE.json's
metric is quoted on ("images/sec (1024x1024, ViT-H)", configs[3]: rsprompter_anchor, SAM ViT-H, 64 tiles over 8
GPUs): 8 x 1024 x 1024 per GPU per step, weak scaling (every rank runs its own batch of 8), so N=1 is the per-GPU
slice of configs[3] and N=8 is configs[3] itself.  `--arch base` gives configs[1], `--model query --arch large
--batch 16` configs[2].  `python bench.py --gpus N` with N > 1 outside torchrun spawns the N ranks itself.
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


def bench_metas(n, kind, lora):
    """configs[4] is quoted on WHU-shape tiles (512-px images resized x2: the mask logits go through the second resize
    of RSMaskFormerFusionHead.predict); every other line uses plain 1024-px metas.  tests/golden/make_golden_bench.py
    builds the canary goldens on the same metas."""
    from rsprompter_amd.synth import synth_metas
    if kind == 'query' and lora:
        return synth_metas(n, ori_shape=(512, 512), scale_factor=(2.0, 2.0))
    return synth_metas(n)


def build_model(arch, num_classes, device, kind='anchor', lora=False):
    import rsprompter_amd as ra
    from rsprompter_amd.default_configs import rsprompter_anchor, rsprompter_query, rsprompter_query_lora
    from rsprompter_amd.synth import synth_state_dict
    if lora and kind != 'query':
        raise SystemExit('bench.py: --lora is the BASELINE.json configs[4] tree (rsprompter_query + LoRA): use --model query')
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        cfg = rsprompter_anchor(arch, num_classes) if kind == 'anchor' else (
            rsprompter_query_lora(arch, num_classes) if lora else rsprompter_query(arch, num_classes))
        model = ra.build_model(cfg)
    model.load_state_dict(synth_state_dict(model, seed=0), strict=True)
    return model.to(device)


def cpu_baseline(arch, num_classes, passes=3, kind='anchor', lora=False):
    """The reference path restated on the CPU (oracle/), timed on this box's host cores: SAME architecture as the
    GPU line, 1 warm-up + `passes` timed runs per stage (mmdet/utils/benchmark.py:208-244), one tile.
    Bounded sample: the ViT encoder is timed as patch-embed + ONE windowed layer + ONE global layer + neck and
    extrapolated by the layer counts (all windowed / all global layers have identical shapes); every other stage
    (feature aggregator, FPN, then RPN, RoI heads, SAM mask decoder, mask post-process on the anchor path / the
    Mask2Former prompter with its SAM decoder call and the fusion head on the query path) runs in full."""
    from oracle import glue
    from oracle.anchor import AnchorOracle
    from oracle import hf_sam
    from rsprompter_amd.synth import synth_images, synth_metas, synth_state_dict
    if kind == 'query':
        return _cpu_baseline_query(arch, num_classes, passes, lora)

    def timeit(fn):
        out = fn()                                   # warm-up
        ts = []
        for _ in range(passes):
            t = time.perf_counter()
            out = fn()
            ts.append(time.perf_counter() - t)
        return sum(ts) / len(ts), out

    with torch.no_grad():
        o = AnchorOracle(arch, num_classes)
        o.load_state_dict(synth_state_dict(o, seed=0))
        enc = o.backbone.vision_encoder
        cfg = hf_sam.ARCH[arch]
        depth, glob = cfg['num_hidden_layers'], list(cfg['global_attn_indexes'])
        win = [i for i in range(depth) if i not in glob]
        n = 1
        metas = synth_metas(n)
        x = glue.data_preprocess(synth_images(n), [123.675, 116.28, 103.53], [58.395, 57.12, 57.375], True, 32)
        first = lambda r: r[0] if isinstance(r, (tuple, list)) else r
        t_patch, h0 = timeit(lambda: enc.patch_embed(x) + enc.pos_embed)
        t_win, h1 = timeit(lambda: first(enc.layers[win[0]](h0)))
        t_glob, h2 = timeit(lambda: first(enc.layers[glob[0]](h1)))
        t_neck, emb = timeit(lambda: enc.neck(h2))
        t_enc = t_patch + len(win) * t_win + len(glob) * t_glob + t_neck
        # downstream stages on hidden states of realistic magnitude (the three tensors above, cycled)
        hidden = tuple([h0] + [(h1, h2)[i % 2] for i in range(depth)])
        G = o.shared_image_embedding.shared_image_embedding.positional_embedding
        ipe = glue.image_wide_positional_embeddings(G, emb.shape[-1]).repeat(emb.shape[0], 1, 1, 1)
        t_agg, feats = timeit(lambda: o.neck.feature_spliter(o.neck.feature_aggregator(hidden)))
        t_rpn, (props, _) = timeit(lambda: o.rpn_predict(feats, metas))
        x_pe = o.add_extra_pe(feats)
        t_box, (dets, _) = timeit(lambda: o.bbox_predict(x_pe, [p['bboxes'] for p in props], metas))
        t_mask, _ = timeit(lambda: o.mask_predict(x_pe, dets, metas, emb, ipe))
        n_det = int(sum(d['bboxes'].shape[0] for d in dets))
        # and ONE untimed-by-stage pass of the whole predict call (every layer really executed): the check on the
        # extrapolated encoder figure
        t = time.perf_counter()
        o.predict(x, metas)
        t_e2e = time.perf_counter() - t
    total = t_enc + t_agg + t_rpn + t_box + t_mask
    return dict(value=round(n / total, 5), unit='images/s', cores=torch.get_num_threads(), kind='port',
                end_to_end_s=round(t_e2e, 3), end_to_end_images_per_s=round(n / t_e2e, 5),
                stages_s=dict(encoder=round(t_enc, 3), encoder_patch=round(t_patch, 3), encoder_window_layer=round(t_win, 3),
                              encoder_global_layer=round(t_glob, 3), encoder_neck=round(t_neck, 3),
                              neck=round(t_agg, 3), rpn=round(t_rpn, 3), bbox_head=round(t_box, 3), mask_head=round(t_mask, 3)),
                sample=f'1 x 1024x1024 tile, CPU oracle (fp32 PyTorch, HF SAM eager attention), SAM-ViT-{arch}; '
                       f'1 warm-up + {passes} timed passes per stage; encoder = patch + {len(win)} x windowed layer + '
                       f'{len(glob)} x global layer + neck from one timed layer of each kind; all other stages in full '
                       f'({n_det} prompt sets through the SAM decoder)')


def _cpu_baseline_query(arch, num_classes, passes, lora):
    from oracle import glue
    from oracle import hf_sam
    from oracle.query import QueryOracle, fusion_predict
    from rsprompter_amd.synth import synth_images, synth_state_dict
    import torch.nn.functional as F

    def timeit(fn):
        out = fn()
        ts = []
        for _ in range(passes):
            t = time.perf_counter()
            out = fn()
            ts.append(time.perf_counter() - t)
        return sum(ts) / len(ts), out

    with torch.no_grad():
        o = QueryOracle(arch, num_classes, 100, max_per_image=100, lora=dict(r=16, alpha=32) if lora else None)
        o.load_state_dict(synth_state_dict(o, seed=0))
        enc = o.backbone.vision_encoder
        if lora:
            enc = enc.base_model.model                    # peft's wrapping (oracle/vitsam.py PeftWrapped)
        cfg = hf_sam.ARCH[arch]
        depth, glob = cfg['num_hidden_layers'], list(cfg['global_attn_indexes'])
        win = [i for i in range(depth) if i not in glob]
        metas = bench_metas(1, 'query', lora)
        x = glue.data_preprocess(synth_images(1), [123.675, 116.28, 103.53], [58.395, 57.12, 57.375], True, 32)
        first = lambda r: r[0] if isinstance(r, (tuple, list)) else r
        t_patch, h0 = timeit(lambda: enc.patch_embed(x) + enc.pos_embed)
        t_win, h1 = timeit(lambda: first(enc.layers[win[0]](h0)))
        t_glob, h2 = timeit(lambda: first(enc.layers[glob[0]](h1)))
        t_neck, emb = timeit(lambda: enc.neck(h2))
        t_enc = t_patch + len(win) * t_win + len(glob) * t_glob + t_neck
        hidden = tuple([h0] + [(h1, h2)[i % 2] for i in range(depth)])
        G = o.shared_image_embedding.shared_image_embedding.positional_embedding
        ipe = glue.image_wide_positional_embeddings(G, emb.shape[-1]).repeat(emb.shape[0], 1, 1, 1)
        t_agg, feats = timeit(lambda: o.neck.feature_spliter(o.neck.feature_aggregator(hidden)))
        t_head, (cls, mask, _) = timeit(lambda: o.panoptic_head(feats, emb, ipe))

        def fuse():
            up = F.interpolate(mask, size=tuple(metas[0]['batch_input_shape'][:2]), mode='bilinear', align_corners=False)
            return fusion_predict(cls, up, metas, num_classes, 100, True)
        t_fuse, _ = timeit(fuse)
        t = time.perf_counter()
        o.predict(x, metas)                                   # one pass of the whole predict call, every layer executed
        t_e2e = time.perf_counter() - t
    total = t_enc + t_agg + t_head + t_fuse
    return dict(value=round(1 / total, 5), unit='images/s', cores=torch.get_num_threads(), kind='port',
                end_to_end_s=round(t_e2e, 3), end_to_end_images_per_s=round(1 / t_e2e, 5),
                stages_s=dict(encoder=round(t_enc, 3), encoder_patch=round(t_patch, 3), encoder_window_layer=round(t_win, 3),
                              encoder_global_layer=round(t_glob, 3), encoder_neck=round(t_neck, 3),
                              neck=round(t_agg, 3), query_head=round(t_head, 3), fusion_head=round(t_fuse, 3)),
                sample=f'1 x 1024x1024 tile, CPU oracle (fp32 PyTorch, HF SAM eager attention), SAM-ViT-{arch}'
                       + (' + LoRA(qkv, r16) unmerged' if lora else '') +
                       f'; 1 warm-up + {passes} timed passes per stage; encoder = patch + {len(win)} x windowed layer + '
                       f'{len(glob)} x global layer + neck from one timed layer of each kind; pixel decoder, masked decoder, '
                       f'SAM mask decoder (100 prompt sets) and fusion head in full')


def canary_name(kind, arch, lora):
    return f'bench_canary_{kind}_{arch}' + ('_lora' if lora else '') + '.pt'


def _parity_canary_query(model, res, out, path):
    """query path: class logits and SAM mask logits of ALL Nq queries of tile 0 (no selection involved).  A query whose
    thresholded attention mask differs from the fp32 oracle's in one bit moves by more than 1e-3 -- for the reference's
    own fp32 forward as well -- so `ok` holds every query to 1e-2 and all but 4 to 1e-3."""
    g = torch.load(path, map_location='cpu', weights_only=True)
    out['golden'] = os.path.relpath(path, ROOT)
    emb = model._last_embeddings
    out['image_embedding_max_abs_err'] = float((emb[0, :, ::8, ::8].float().cpu() - g['embedding_sample']).abs().max())
    cls, lazy = model._last_head_out
    low = lazy.low_res.detach().float().cpu()                 # [B, Nq, 256, 256]
    per_q = (low[0, :, ::16, ::16] - g['low_res_sample']).abs().flatten(1).amax(1)
    out['class_logit_max_abs_err'] = float((cls[0].detach().float().cpu() - g['cls_pred']).abs().max())
    out['mask_logit_max_abs_err'] = float(per_q.max())
    out['mask_logit_err_5th_largest'] = float(per_q.sort(descending=True).values[min(4, per_q.numel() - 1)])
    out['queries_within_1e-3'] = f'{int((per_q < 1e-3).sum())}/{per_q.numel()}'
    same = res[0].pred_instances.query_indices.cpu().long() == g['query_indices']
    out['query_indices_equal'] = f'{int(same.sum())}/{same.numel()}'
    out['mask_logit_range'] = g['low_res_absmax']
    out['tolerance'] = 1e-3
    out['ok'] = bool(out['finite'] and out['image_embedding_max_abs_err'] < 1e-3 and out['class_logit_max_abs_err'] < 1e-3
                     and int((per_q < 1e-3).sum()) >= per_q.numel() - 4 and float(per_q.max()) < 1e-2
                     and int(same.sum()) >= same.numel() - 4)
    return out


def parity_canary(model, imgs, metas, arch, kind, lora=False, res=None):
    """One extra step OUTSIDE the timed region on the bench's own inputs, tile 0 compared with the CPU oracle's answer
    for exactly this fixture (tests/golden/bench_canary_anchor_<arch>.pt, written by tests/golden/make_golden_bench.py;
    nothing under oracle/ is imported here).  Reports whether every output of the step is finite, the max-abs error
    of the image embedding and of the SAM low-resolution mask logits of the detections matched to the oracle's, and how
    many matched.  A bench that runs on NaN rows, stale kernels or a wrong weight layout says so on its own line.
    res: the outputs of a test_step the caller has just run on these inputs with rsprompter_amd.debug.KEEP_TRACES on (the GPU
    tests hold tile 0 of the bench fixtures against the same goldens); None: run the step here."""
    import rsprompter_amd.debug as dbg
    from rsprompter_amd.structures import DetDataSample
    out = dict(finite=None, golden=None)
    keep, dbg.KEEP_TRACES = dbg.KEEP_TRACES, True
    try:
        if res is None:
            res = model.test_step(dict(inputs=imgs, data_samples=[DetDataSample(metainfo=dict(m)) for m in metas]))
        torch.cuda.synchronize()
        fin = True
        for r in res:
            pi = r.pred_instances
            fin &= bool(torch.isfinite(pi.bboxes).all()) and bool(torch.isfinite(pi.scores).all())
        emb = getattr(model, '_last_embeddings', None)
        if emb is not None:
            fin &= bool(torch.isfinite(emb).all())
        low = None
        if kind == 'anchor':
            tr = model.roi_head._last_mask_trace
            low = None if tr is None else tr['mask_preds']
            if low is not None:
                fin &= bool(torch.isfinite(low).all())
        out['finite'] = bool(fin)
        path = os.path.join(ROOT, 'tests', 'golden', canary_name(kind, arch, lora))
        if kind == 'query' and os.path.exists(path):
            return _parity_canary_query(model, res, out, path)
        if kind != 'anchor' or not os.path.exists(path) or low is None:
            return out
        g = torch.load(path, map_location='cpu', weights_only=True)
        out['golden'] = os.path.relpath(path, ROOT)
        out['image_embedding_max_abs_err'] = float((emb[0, :, ::8, ::8].float().cpu() - g['embedding_sample']).abs().max())
        pi = res[0].pred_instances
        k0 = int(pi.labels.shape[0])
        pb, ps, pl = pi.bboxes.float().cpu(), pi.scores.float().cpu(), pi.labels.cpu()
        sample = low[:k0, 0, ::16, ::16].float().cpu()
        used, errs, same_idx, rank_same = set(), [], 0, 0
        pc = pi.cand_index.cpu().long() if hasattr(pi, 'cand_index') else None
        for j in range(g['labels'].shape[0]):          # same label, same box (1e-2 px), same score (1e-4): tests/_match.py
            d = (pb - g['bboxes'][j]).abs().amax(1)
            d[(pl != g['labels'][j])] = float('inf')
            for u in used:
                d[u] = float('inf')
            i = int(d.argmin()) if d.numel() else -1
            if i >= 0 and float(d[i]) < 1e-2 and abs(float(ps[i]) - float(g['scores'][j])) < 1e-4:
                used.add(i)
                rank_same += int(i == j)
                errs.append(float((sample[i] - g['low_res_sample'][j]).abs().max()))
        out['detections_matched'] = f'{len(errs)}/{int(g["labels"].shape[0])}'
        # ... of which at the oracle's own rank (the same box, score and label in the same output row)
        out['ranks_equal'] = f'{rank_same}/{int(g["labels"].shape[0])}'
        if pc is not None and 'cand' in g:
            # index equality, position by position: detection j of the tile is the oracle's detection j -- the same kept
            # candidate (proposal x class entry of the free-running R-CNN stage) with the same label
            n = min(k0, int(g['cand'].shape[0]))
            same_idx = int(((pc[:n] == g['cand'][:n]) & (pl[:n] == g['labels'][:n])).sum())
            out['indices_equal'] = f'{same_idx}/{int(g["cand"].shape[0])}'
            diff = (~((pc[:n] == g['cand'][:n]) & (pl[:n] == g['labels'][:n]))).nonzero()[:, 0].tolist()
            # rows that differ: [row, this run's candidate index, the oracle's] (at most 4 listed)
            out['indices_unequal_rows'] = [[j, int(pc[j]), int(g['cand'][j])] for j in diff[:4]]
        out['mask_logit_max_abs_err'] = max(errs) if errs else None
        out['mask_logit_range'] = g['low_res_absmax']
        out['tolerance'] = 1e-3
        out['ok'] = bool(fin and errs and max(errs) < 1e-3 and out['image_embedding_max_abs_err'] < 1e-3
                         and len(errs) >= int(g['labels'].shape[0]) - 4)
    finally:
        dbg.KEEP_TRACES = keep
    return out


def _pmc_traffic(arch):
    """HBM bytes per launch of the dominant GEMM from the committed rocprofv3 PMC passes (profiles/r1_pmc/,
    FETCH_SIZE x2 + WRITE_SIZE as MI355X_MICROARCH.md prescribes); the counters cannot be collected inside this
    process, so the number is the one measured with tools/pmc_round2.sh on the kernel's most expensive shape of this
    architecture; None when no pass exists for it."""
    root = os.path.dirname(os.path.abspath(__file__))
    for rel in (f'profiles/r6_pmc/gemm_traffic_{arch}.json', f'profiles/r5_pmc/gemm_traffic_{arch}.json', f'profiles/r4_pmc/gemm_traffic_{arch}.json', f'profiles/r3_pmc/gemm_traffic_{arch}.json', f'profiles/r2_pmc/gemm_traffic_{arch}.json', 'profiles/r1_pmc/gemm_lin1_traffic.json'):
        try:
            with open(os.path.join(root, rel)) as fh:
                t = json.load(fh)
            if arch != 'base' and rel.startswith('profiles/r1_pmc'):
                return None                      # that pass was taken on the ViT-B shape only
            return {'bytes_per_launch': t['traffic_bytes_per_launch'], 'algorithmic_bytes_per_launch': t['algorithmic_bytes_per_launch'],
                    'shape': t['shape'], 'source': rel, 'measured_in_this_run': False,
                    'how': 'rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of this kernel shape on another MI355X box, committed under '
                           'profiles/; PMC counters cannot be read from inside the benchmark process'}
        except Exception:
            continue
    return None


def _config_tag(args, B, world):
    if args.model == 'anchor' and args.arch == 'huge' and B == 8:
        return (' (BASELINE.json configs[3]: 64 tiles sharded over 8 GPUs)' if world == 8 else
                f' (the per-GPU slice of BASELINE.json configs[3] on {world} GPU' + ('s)' if world > 1 else ')'))
    if args.model == 'anchor' and args.arch == 'base' and B == 8 and world == 1:
        return ' (BASELINE.json configs[1])'
    if args.model == 'query' and args.arch == 'large' and B == 16 and world == 1:
        return ' (BASELINE.json configs[2])'
    if args.model == 'query' and args.arch == 'huge' and B == 4 and args.lora:
        return (' (BASELINE.json configs[4]: batch 32 over 8 GPUs, WHU-shape metas)' if world == 8 else
                f' (the per-GPU slice of BASELINE.json configs[4] on {world} GPU' + ('s' if world > 1 else '') + ', WHU-shape metas)')
    return ''


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--arch', default='huge', choices=['base', 'large', 'huge'])
    ap.add_argument('--batch', type=int, default=8, help='images per GPU per step')
    ap.add_argument('--model', default='anchor', choices=['anchor', 'query'],
                    help="prompter variant; the headline line is 'anchor' (BASELINE.json configs[3])")
    ap.add_argument('--lora', action='store_true',
                    help='--model query only: LoRA(qkv, r16, alpha32) adapters on the encoder and WHU-shape metas '
                         '(BASELINE.json configs[4]: --model query --arch huge --batch 4 --lora)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--shapes', action='store_true', help='break the kernel table down by GEMM/attention shape')
    ap.add_argument('--host-inputs', action='store_true',
                    help='NOT the headline: the tiles start in pinned host memory every step (PCIe-inclusive rate for DESIGN.md)')
    ap.add_argument('--f8corr', nargs='?', const='all', default=None, choices=['all', 'mlp'],
                    help='opt-in fast mode: encoder GEMMs as fp16 hi.hi + one fp8 correction MFMA (DESIGN.md section 3); '
                         'NOT the headline configuration -- parity margins are 8x smaller')
    ap.add_argument('--exchange', choices=['gather', 'allgather'], default='allgather',
                    help="N > 1: where the per-image results of a step go -- 'allgather' (default since round 6): every rank, "
                         "BASELINE.json north_star's \"RCCL all-gather of instance results\"; 'gather': rank 0 only (mmengine "
                         "collect_results, what tools/dist_test.sh evaluates)")
    args = ap.parse_args()

    if not torch.cuda.is_available():
        raise RuntimeError('bench.py needs an MI355X (no CPU fallback)')
    if args.gpus > 1 and 'RANK' not in os.environ:
        # plain `python bench.py --gpus N`: become the launcher of N ranks (one process per GPU, RCCL), same flags
        have = torch.cuda.device_count()
        if have < args.gpus:
            raise SystemExit(f'bench.py: --gpus {args.gpus} requested but only {have} GPU(s) are visible')
        import socket
        import subprocess
        sk = socket.socket()
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
        sk.close()
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}',
               '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))

    from rsprompter_amd import dist as rdist
    from rsprompter_amd import ops
    if args.f8corr:
        ops.F8_CORR = True if args.f8corr == 'all' else args.f8corr
    from rsprompter_amd.structures import DetDataSample
    from rsprompter_amd.synth import synth_images, synth_metas
    import torch.distributed as tdist

    rank, local, world = rdist.init_from_env()
    if world != args.gpus:
        raise SystemExit(f'bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with '
                         f'--nproc-per-node {args.gpus} (or run plain `python bench.py --gpus {args.gpus}`)')
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    num_classes = 10 if args.model == 'anchor' else 1
    model = build_model(args.arch, num_classes, dev, args.model, args.lora)
    B = args.batch
    host_imgs = [im.pin_memory() for im in synth_images(B, seed=1234 + 1000 * rank)]
    imgs = [im.to(dev) for im in host_imgs]
    metas = bench_metas(B, args.model, args.lora)

    # N > 1: the result exchange of step i (records + COCO RLE strings of every instance, produced by device kernels and
    # gathered to rank 0 like mmengine collect_results: rsprompter_amd/dist.py::gather_results) is queued completely --
    # kernels, collectives, pinned-host copy -- on a side stream when the step ends; the host only waits for that stream's
    # event after step i + 1 has been launched and receives a lazy view (no per-instance Python work on any rank).
    # Every exchange is collected inside the timed region (sync() drains the last).
    side = torch.cuda.Stream(device=dev) if world > 1 else None
    pending = [None]

    def step():
        samples = [DetDataSample(metainfo=dict(m)) for m in metas]
        step_imgs = [im.to(dev, non_blocking=True) for im in host_imgs] if args.host_inputs else imgs
        out = model.test_step(dict(inputs=step_imgs, data_samples=samples))
        res = [o.pred_instances for o in out]
        if world > 1:
            if pending[0] is not None:
                pending[0].collect()
            pending[0] = rdist.gather_results(res, dataset_size=world * B, stream=side, dst=0 if args.exchange == 'gather' else None)
        return res

    def sync():
        if pending[0] is not None:
            pending[0].collect()
            pending[0] = None
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
    sync()
    canary = parity_canary(model, imgs, metas, args.arch, args.model, args.lora) if rank == 0 else None

    result = None
    # ---- roofline leg: one extra instrumented step, per-kernel HIP events on the launch stream.  Every rank runs
    # the step (it contains the result gather); only rank 0 records and reports ----
    prof = ops.Profiler() if rank == 0 else None
    if prof is not None:
        prof.shapes = args.shapes
        ops.set_profiler(prof)
    step()
    sync()
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
        attn = [v for k, v in agg.items() if k.startswith('attn_stream_kernel<vit') or k.startswith('attn_kernel<vit') or
                k.startswith('attn_global_kernel') or k.startswith('attn_win_kernel<vit')]
        # the windowed layers' FLOPs are counted for the padded 5 x 5 x 196 tokens of SURVEY section 8(d); the kernel
        # evaluates only the 64 x 64 real tokens as queries (all 196 keys each): 4096 / 4900 of that figure
        win_fl = sum(v['flops'] for k, v in agg.items() if k.startswith('attn_win_kernel<vit'))
        attn_fl_eval = sum(v['flops'] for v in attn) - win_fl * (1.0 - 4096.0 / 4900.0)
        attn_ms = sum(v['ms'] for v in attn)
        attn_tf = sum(v['flops'] for v in attn) / attn_ms / 1e9 if attn_ms else None
        relpos_ms = sum(v['ms'] for k, v in agg.items() if k.startswith('vit_relpos'))
        attn_tf_rel = sum(v['flops'] for v in attn) / (attn_ms + relpos_ms) / 1e9 if attn_ms else None
        value = world * B * args.steps / elapsed
        result = {
            'metric': 'images/sec (1024x1024 synthetic tiles, rsprompter_%s SAM-ViT-%s%s, full predict path)' % (
                args.model, args.arch[0].upper(), ' + LoRA' if args.lora else ''),
            'value': round(value, 3), 'unit': 'images/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(elapsed / args.steps * 1e3, 3), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None,
            'dtype': 'f32 (fp32 in/out; GEMMs and attention as fp16x3 split-precision MFMA with fp32 accumulate'
                     + ('; --f8corr: encoder GEMMs as fp16 hi.hi + one fp8 (e4m3, MX block scales) correction MFMA)' if args.f8corr else ')'),
            'data': 'synthetic' + (' (tiles copied from pinned host memory every step: not the headline)' if args.host_inputs else ''),
            'config': {'workload': f'rsprompter_{args.model} SAM-ViT-{args.arch}' + (' + LoRA(qkv r16)' if args.lora else '') + f', batch {B}x1024x1024 per GPU, '
                                   f'{num_classes} classes, seeded synthetic weights' + _config_tag(args, B, world),
                       'images_per_gpu_per_step': B, 'detections_per_step_rank0': n_dets,
                       'parallelism': (f'dp{world} (images sharded by batch, result ' + ('gather to rank 0' if args.exchange == 'gather' else 'all-gather') + ' over RCCL)') if world > 1 else 'single GPU',
                       },
            'roofline': {'bound': 'mfma', 'kernel': dom_name, 'launches_per_step': dom['calls'],
                         'ms_per_step': round(dom['ms'], 3), 'achieved': round(achieved, 2),
                         'peak': PEAK_F16_MFMA_TFLOPS, 'unit': 'TFLOP/s', 'frac': round(achieved / PEAK_F16_MFMA_TFLOPS, 4),
                         'traffic': (_pmc_traffic(args.arch) or {}).get('bytes_per_launch'), 'traffic_detail': _pmc_traffic(args.arch),
                         'traffic_measured_in_this_run': False,
                         'note': 'achieved = algorithmic fp32 FLOPs (2MNK) of all launches of the kernel in one step / '
                                 'their summed HIP-event durations; the kernel issues 3 fp16 MFMA passes per algorithmic '
                                 'FLOP (fp16x3), so its ceiling is peak/3 = 833 TFLOP/s'
                                 + (' (gemm_f16f8: 2 fp16 + 1 fp8 K=64 MFMA per 32 k = 2 units of matrix time, ceiling peak/2)' if args.f8corr else ''),
                         'frac_of_fp16x3_ceiling': round(achieved / (PEAK_F16_MFMA_TFLOPS / 3), 4)},
            'roofline_attention': {'bound': 'mfma', 'kernel': 'attn_win_kernel (windowed layers, rel-pos inside) + attn_stream_kernel (global layers)',
                                   'achieved': None if attn_tf is None else round(attn_tf, 2),
                                   'achieved_evaluated_queries': None if not attn_ms else round(attn_fl_eval / attn_ms / 1e9, 2),
                                   'achieved_incl_relpos_kernels': None if attn_tf_rel is None else round(attn_tf_rel, 2),
                                   'ms_per_step': round(attn_ms, 3), 'relpos_ms_per_step': round(relpos_ms, 3),
                                   'peak': PEAK_F16_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                                   'frac': None if attn_tf is None else round(attn_tf / PEAK_F16_MFMA_TFLOPS, 4),
                                   'attention_gemm_gflop_per_image': ATTN_GEMM_GFLOP_PER_IMAGE[args.arch]},
            'exchange': None if world == 1 else ('all-gather of instance results (records + COCO RLE strings) to every rank' if args.exchange == 'allgather'
                                                 else 'gather of instance results to rank 0'),
            'rccl_ranks': world if world > 1 else 0,
            'parity_canary': canary,
            'kernels': kernels,
        }
        if world == 1 and not args.no_cpu_baseline:
            try:
                result['cpu_baseline'] = cpu_baseline(args.arch, num_classes, kind=args.model, lora=args.lora)
            except Exception as e:  # the baseline is context, never a reason to lose the measurement
                result['cpu_baseline'] = {'value': None, 'unit': 'images/s', 'cores': torch.get_num_threads(),
                                          'kind': 'port', 'sample': f'failed: {e!r}'}
        print(json.dumps(result), flush=True)
    if world > 1:
        tdist.barrier()
        tdist.destroy_process_group()


if __name__ == '__main__':
    main()
