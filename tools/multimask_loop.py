"""Round 6: hunt for the box-dependent answer of the multimask decoder chain ("weak" 1).

    python tools/multimask_loop.py [--iters 200] [--hw 64] [--multimask 1] [--fused 0|1]

Runs RSPrompterAnchorMaskHead on one seeded input `iters` times; every torch.empty of an iteration starts as a byte
pattern that rotates (0xFF = NaN, 0x00, 0x7B = large finite fp16/fp32, nothing = the allocator's own stale contents after
torch.cuda.empty_cache()).  Every iteration's stage tensors (tokens, per-RoI key planes, ConvTranspose + LN + GELU planes,
hyper-network vectors, masks, iou) are held against the HF decoder (computed once on the CPU) AND bit for bit against the
first iteration: a race or an uninitialised read shows as the first stage whose bits move.  Test infrastructure."""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--iters', type=int, default=200)
    ap.add_argument('--hw', type=int, default=64)
    ap.add_argument('--multimask', type=int, default=1)
    ap.add_argument('--fused', type=int, default=1, help='0: the kernel chains instead of t2i_fold / upscale_fused')
    ap.add_argument('--rois', type=int, default=7)
    ap.add_argument('--empty-cache', type=int, default=1)
    a = ap.parse_args()
    from oracle import hf_sam
    from rsprompter_amd.registry import MODELS
    from rsprompter_amd.synth import synth_state_dict
    from test_gpu_baseline_configs import _hf_decoder_stages, _planes_f32
    emu = os.environ.get('RSP_WAVE_EMU') == '1'           # developer check of this script on the CPU emulator (small --hw)
    if emu:
        sys.path.insert(0, os.path.join(ROOT, 'tests', 'wave_emu'))
        import harness
        ctx = harness.emulated_ops()
        ctx.__enter__()
        torch.cuda.empty_cache = lambda: None
    dev = torch.device('cpu' if emu else 'cuda:0')
    hw, R, B = a.hw, a.rois, 2
    mm = bool(a.multimask)
    head = MODELS.build(dict(type='RSPrompterAnchorMaskHead', mask_decoder=dict(type='RSSamMaskDecoder', hf_pretrain_name='sam_vit_base'),
                             in_channels=256, roi_feat_size=14, per_pointset_point=5, with_sincos=True, multimask_output=mm,
                             class_agnostic=True))
    sd = synth_state_dict(head, 3)
    head.load_state_dict(sd)
    head = head.to(dev)
    dec = hf_sam.build_mask_decoder()
    dec.load_state_dict({k[len('mask_decoder.mask_decoder.'):]: v for k, v in sd.items() if k.startswith('mask_decoder.mask_decoder.')})
    g = torch.Generator().manual_seed(0)
    x = torch.randn(R, 256, 14, 14, generator=g)
    emb = torch.randn(B, 256, hw, hw, generator=g)
    ipe = torch.randn(1, 256, hw, hw, generator=g).expand(B, -1, -1, -1).contiguous()
    roi_img = (torch.arange(R) * B // R).to(torch.int64)
    cl = lambda t: t.to(dev).contiguous(memory_format=torch.channels_last)
    hip = head.mask_decoder.mask_decoder
    hip.keep_stages = True
    hip.t2i_fold = hip.upscale_fused = bool(a.fused)
    xs, es, ps, ri = cl(x), cl(emb), cl(ipe), roi_img.to(dev)
    sparse = head.point_embeddings(xs).cpu()
    nm = 3 if mm else 1
    (ref_m, ref_i, *_), ref = _hf_decoder_stages(
        dec, image_embeddings=emb[roi_img], image_positional_embeddings=ipe[roi_img], sparse_prompt_embeddings=sparse.unsqueeze(1),
        dense_prompt_embeddings=sd['no_mask_embed.weight'].reshape(1, -1, 1, 1).expand(R, -1, hw, hw), multimask_output=mm)
    T = 1 + 4 + 5
    refs = dict(tokens=ref['tokens'].reshape(R, T, 256), keys=ref['keys'].reshape(R * hw * hw, 256),
                up=ref['up'].permute(0, 2, 3, 1).reshape(-1, 64), masks=ref_m.reshape(R, nm, 4 * hw, 4 * hw), iou=ref_i.reshape(R, nm))
    toks = (1, 2, 3) if mm else (0,)
    for n, i in enumerate(toks):
        refs[f'hyper{n}'] = ref[f'hyper{i}'].reshape(R, 32)

    pattern = [None]
    e, el = torch.empty, torch.empty_like

    def fill(t):
        if pattern[0] is not None and t.numel() and t.is_contiguous():
            t.view(torch.uint8).fill_(pattern[0])
        return t
    torch.empty = lambda *aa, **k: fill(e(*aa, **k))
    torch.empty_like = lambda *aa, **k: fill(el(*aa, **k))

    def run():
        low, iou = head(xs, es, ps, ri)
        st = hip._last_stages
        out = dict(tokens=st['tokens'], keys_hi=st['keys'].hi, keys_lo=st['keys'].lo, masks=low, iou=iou)
        if st['up'] is not None:
            out['up_hi'], out['up_lo'] = st['up'].hi, st['up'].lo
        for n, h in enumerate(st['hyper']):
            out[f'hyper{n}'] = h
        return out

    def to_ref_form(k, dct):
        if k == 'keys_hi':
            return _planes_f32(hip._last_stages['keys']), refs['keys']
        if k == 'up_hi':
            return _planes_f32(hip._last_stages['up']), refs['up']
        if k in refs:
            return dct[k].cpu(), refs[k]
        return None, None

    pattern[0] = None
    first = {k: v.clone() for k, v in run().items()}
    for k in first:
        got, want = to_ref_form(k, first)
        if got is not None:
            print(f'iteration 0: {k} err {float((got - want).abs().max()):.2e}', flush=True)
    bad = wrong = n_wrong_vals = n_hist = 0
    wrong += float((first['masks'].cpu() - refs['masks']).abs().max()) > 1e-3
    pats = [0xFF, 0x00, 0x7B, None]
    for it in range(1, a.iters):
        pattern[0] = pats[it % 4]
        if pattern[0] is None and a.empty_cache:
            torch.cuda.empty_cache()
        out = run()
        if os.environ.get('RSP_LOOP_SELFTEST') == '1':          # developer check of the analysis below: drop one addend by hand
            import torch.nn.functional as F
            with torch.no_grad():
                y_ = F.gelu(dec.upscale_conv2(ref['up']))
            out['masks'][1, 0, 5, 34] -= float(y_[1, 13, 5, 34] * refs['hyper0'][1, 13])
        else:
            torch.cuda.synchronize()
        moved = [k for k, v in out.items() if not torch.equal(v, first[k])]
        e_ref = float((out['masks'].cpu() - refs['masks']).abs().max())
        wrong += e_ref > 1e-3                        # against the HF reference: independent of what iteration 0 produced
        if e_ref > 1e-3:
            # WHERE, in the terms of sam_upscale2_kernel: sub-pixel j = 2 (Y & 1) + (X & 1) of output (Y, X), lane quarter of
            # the input pixel x = X >> 1 inside its 32-pixel wave tile, mask token
            w = ((out['masks'].cpu() - refs['masks']).abs() > 1e-3)
            n_wrong_vals += int(w.sum())
            if n_hist < 3:
                n_hist += 1
                idx = w.nonzero()
                Y, X = idx[:, -2], idx[:, -1]
                j = (2 * (Y & 1) + (X & 1)).bincount(minlength=4).tolist()
                q = (((X >> 1) & 31) >> 4).bincount(minlength=2).tolist()
                tok = idx[:, 1].bincount(minlength=w.shape[1]).tolist() if w.dim() == 4 else None
                print(f'iteration {it}: {int(w.sum())} mask values off the HF reference by > 1e-3: by sub-pixel j {j}, by lane quarter '
                      f'(pixels 0-15 / 16-31 of the wave tile) {q}, by mask token {tok}, by RoI {idx[:, 0].bincount(minlength=w.shape[0]).tolist()}',
                      flush=True)
                if w.dim() == 4 and 'up' in ref:
                    # WHAT the wrong value is, in terms of the kernel's sum: the output is sum_c GELU(ConvT2(up)[c]) hyper[c] over 32
                    # channels, 16 in the storing half wave (channels 8 g + 4 hh + e, hh = Y & 1) and 16 from its partner; which
                    # addends are missing from the value that was stored?
                    import torch.nn.functional as F
                    with torch.no_grad():
                        y = F.gelu(dec.upscale_conv2(ref['up']))                       # [R, 32, 4h, 4w]
                    from collections import Counter
                    votes = Counter()
                    for (r, t, Y_, X_) in idx[:48].tolist():
                        pc = y[r, :, Y_, X_] * refs[f'hyper{t}'][r]                  # the 32 addends
                        got, tot = float(out['masks'][r, t, Y_, X_]), float(pc.sum())
                        hh_ = Y_ & 1
                        own = [8 * g_ + 4 * hh_ + e_ for g_ in range(4) for e_ in range(4)]
                        oth = [c + 4 - 8 * hh_ for c in own]
                        cands = {'own half only': float(pc[own].sum()), 'partner half only': float(pc[oth].sum())}
                        for c in range(32):
                            cands[f'all but channel {c} ({"own" if c in own else "partner"} half, position {(own if c in own else oth).index(c)})'] = tot - float(pc[c])
                        for g_ in range(4):
                            cands[f'all but own group {g_}'] = tot - float(pc[own[4 * g_:4 * g_ + 4]].sum())
                            cands[f'all but partner group {g_}'] = tot - float(pc[oth[4 * g_:4 * g_ + 4]].sum())
                            cands[f'own groups 0..{g_} + partner'] = float(pc[oth].sum()) + float(pc[own[:4 * g_ + 4]].sum())
                            cands[f'partner groups 0..{g_} + own'] = float(pc[own].sum()) + float(pc[oth[:4 * g_ + 4]].sum())
                        best = min(cands, key=lambda k_: abs(cands[k_] - got))
                        votes[best if abs(cands[best] - got) < 2e-4 else 'none of the candidates'] += 1
                    print(f'  what the first {min(48, idx.shape[0])} wrong values equal: {dict(votes)}', flush=True)
        if not moved:
            continue
        bad += 1
        if bad > 12:
            continue
        for k in moved:
            v = out[k]
            c1, c2 = v.cpu(), v.cpu()                       # two copies: is the DEVICE tensor wrong, or one copy of it?
            again = torch.equal(v, first[k])                 # ... and a second comparison on the device
            f = first[k].cpu()
            d = (c1 != f) | (torch.isnan(c1) != torch.isnan(f))
            idx = d.nonzero()
            lo, hi_ = idx.min(0).values.tolist(), idx.max(0).values.tolist()
            print(f'iteration {it} (pattern {pattern[0]}): {k} shape {tuple(v.shape)} dtype {v.dtype}: {int(d.sum())} values moved, index box '
                  f'{lo} .. {hi_}; second device comparison equal={again}; two host copies equal={torch.equal(c1, c2)}; '
                  f'got {c1[d][:6].tolist()} first {f[d][:6].tolist()}; flat offsets {(d.flatten().nonzero().flatten()[:4]).tolist()} '
                  f'data_ptr {v.data_ptr():#x} nan {int(torch.isnan(c1.float()).sum())}', flush=True)
    print(f'multimask={mm} fused={a.fused} hw={hw} R={R}: {bad} of {a.iters} iterations differ from the first; '
          f'{wrong} of {a.iters} off the HF reference by more than 1e-3 (iteration 0 included), {n_wrong_vals} wrong values in all',
          flush=True)


if __name__ == '__main__':
    main()
