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
    if os.environ.get('RSP_POISON_EMPTY') == '1':
        _poison_empty()


def _poison_empty():
    """RSP_POISON_EMPTY=1 (tools/gpu_job.sh step `n:`): every torch.empty / empty_like / new_empty of the session starts as
    0xFF bytes -- NaN as fp32 / fp16, -1 as an index -- instead of whatever the allocator's block held.  A kernel that reads
    an element nobody wrote then yields NaN instead of a box-dependent answer (round 5's red driver run: same code, same
    seeds, another box's memory).  Test infrastructure; the package allocates through the same three calls."""
    import torch
    if getattr(torch, '_rsp_poisoned', False):
        return
    e, el, ne = torch.empty, torch.empty_like, torch.Tensor.new_empty

    def fill(t):
        if t.numel() and t.is_contiguous() and t.dtype in (torch.float32, torch.float16, torch.int32, torch.int64, torch.uint8,
                                                             torch.int16, torch.bfloat16, torch.float64, torch.int8):
            t.view(torch.uint8).fill_(0xFF)
        return t
    torch.empty = lambda *a, **k: fill(e(*a, **k))
    torch.empty_like = lambda *a, **k: fill(el(*a, **k))
    torch.Tensor.new_empty = lambda self, *a, **k: fill(ne(self, *a, **k))
    torch._rsp_poisoned = True


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


# GPU suite order (round 6; VERDICT r5 "weak" 2): the driver runs `pytest -x`, and alphabetical collection put the 25-s-per-tile
# CPU-oracle end-to-end tests first and the 110 kernel-level tests last -- one late failure hid 147 tests.  Cheap and broad
# first: kernel tier, then the stage / sibling-model / API tests, the end-to-end BASELINE configurations last.
_GPU_FILE_ORDER = ['test_gpu_kernels', 'test_gpu_gemm_s2', 'test_gpu_query', 'test_gpu_dist', 'test_gpu_encoder', 'test_gpu_f8corr',
                   'test_gpu_apis', 'test_gpu_samseg', 'test_gpu_samdet', 'test_gpu_anchor', 'test_gpu_baseline_configs']


def pytest_collection_modifyitems(config, items):
    rank = {name: i for i, name in enumerate(_GPU_FILE_ORDER)}

    def key(item):
        mod = os.path.splitext(os.path.basename(str(item.fspath)))[0]
        return rank.get(mod, -1)              # CPU files keep their place in front (stable sort)
    items.sort(key=key)


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
