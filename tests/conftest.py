import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    config.addinivalue_line('markers', 'quick: kernel-level GPU tests (-m "gpu and quick": the tier run between performance experiments)')
    # the parity tests read intermediate tensors the predict path only keeps on request (rsprompter_amd/debug.py)
    import rsprompter_amd.debug as dbg
    dbg.KEEP_TRACES = True
    _memoise_synth_tensors()


def _memoise_synth_tensors(cap_bytes=12 << 30):
    """The seeded synthetic weights are pure functions of (seed, key, shape, dtype) (rsprompter_amd/synth.py); the suite
    builds the same ViT-H / ViT-L state dicts many times (oracle and product side of every end-to-end test: ~4 s of
    torch.randn per ViT-H tree).  Memoised for the session; load_state_dict copies, nobody writes into the cached tensors."""
    import rsprompter_amd.synth as synth
    if getattr(synth, '_memo_installed', False):
        return
    cache, used = {}, [0]
    plain = synth.synth_tensor

    def cached(name, ref, seed=0):
        key = (seed, name, tuple(ref.shape), str(ref.dtype))
        t = cache.get(key)
        if t is None:
            t = plain(name, ref, seed)
            nb = t.numel() * t.element_size()
            if used[0] + nb <= cap_bytes:
                cache[key] = t
                used[0] += nb
        return t
    synth.synth_tensor = cached
    synth._memo_installed = True


@pytest.fixture(scope='session')
def dev():
    import torch
    if os.environ.get('RSP_WAVE_EMU') == '1':
        # developer mode (tests/wave_emu/README.md): `RSP_WAVE_EMU=1 pytest tests/test_gpu_kernels.py -m gpu -k ...` runs
        # GPU tests on CPU tensors through the lane-level emulation of the kernels -- slow, small shapes only
        sys.path.insert(0, os.path.join(ROOT, 'tests', 'wave_emu'))
        import harness
        with harness.emulated_ops():
            yield torch.device('cpu')
        return
    if not torch.cuda.is_available():
        pytest.skip('no GPU')
    yield torch.device('cuda:0')
