"""Session memo for CPU oracle runs that several GPU test files need on the SAME fixture (round 6: the GPU suite runs
against the driver's wall-clock limit; the ViT-B anchor oracle on two tiles costs ~20 s per run).  Test infrastructure."""
_MEMO = {}


def anchor_base_two_tiles():
    """AnchorOracle('base', 10), weight seed 0, synth_images(2) / synth_metas(2): (oracle, state dict, imgs, metas, x, results, trace)"""
    if 'anchor_base_2' not in _MEMO:
        from oracle import glue
        from oracle.anchor import AnchorOracle
        from rsprompter_amd.synth import synth_images, synth_metas, synth_state_dict
        oracle = AnchorOracle('base', 10)
        sd = synth_state_dict(oracle, seed=0)
        oracle.load_state_dict(sd)
        imgs, metas = synth_images(2), synth_metas(2)
        x = glue.data_preprocess(imgs, [123.675, 116.28, 103.53], [58.395, 57.12, 57.375], True, 32)
        results, trace = oracle.predict(x, metas)
        _MEMO['anchor_base_2'] = (oracle, sd, imgs, metas, x, results, trace)
    return _MEMO['anchor_base_2']
