import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    # the parity tests read intermediate tensors the predict path only keeps on request (rsprompter_amd/debug.py)
    import rsprompter_amd.debug as dbg
    dbg.KEEP_TRACES = True


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
