"""TEST INFRASTRUCTURE: run `rsprompter_amd.ops` wrappers on CPU tensors through the lane-level emulation of the kernels
(tests/wave_emu/emu_hip.h).  Used by tests/test_wave_emu_cpu.py only -- a pytest fixture swaps the library handle, the
stream getter and the device checks of `ops` for the duration of one test and restores them afterwards; nothing in the
package can reach this module."""
import contextlib
import ctypes
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

_EMU = None


def load_emu():
    global _EMU
    if _EMU is None:
        import build
        from rsprompter_amd import _lib
        srcs = sorted(f for f in os.listdir(build.CSRC) if f.endswith('.hip'))
        lib = ctypes.CDLL(build.build(srcs))
        for name, (res, args) in _lib.PROTOTYPES.items():
            fn = getattr(lib, name)        # every symbol of include/rsp_hip.h must exist in the emulated build too
            fn.restype, fn.argtypes = res, args
        lib.emu_set_lazy_dma.argtypes = [ctypes.c_int]
        lib.emu_set_poison_lds.argtypes = [ctypes.c_int]
        if os.environ.get('RSP_WAVE_EMU_COUNT'):
            lib = _Counting(lib, os.environ['RSP_WAVE_EMU_COUNT'])
        _EMU = lib
    return _EMU


class _Counting:
    """RSP_WAVE_EMU_COUNT=<file>: calls per C entry point, appended to <file> as `pid name count` lines when the process
    ends (which entry points of include/rsp_hip.h does the emulator suite execute: profiles/r4_emu_entry_point_coverage.txt)"""

    def __init__(self, lib, path):
        import atexit
        self.__dict__['_lib'], self.__dict__['_n'] = lib, {}
        atexit.register(lambda: open(path, 'a').writelines(f'{os.getpid()} {k} {v}\n' for k, v in sorted(self._n.items())))

    def __getattr__(self, name):
        fn = getattr(self._lib, name)
        if not name.startswith('rsp_'):
            return fn

        def call(*a):
            self._n[name] = self._n.get(name, 0) + 1
            return fn(*a)
        self.__dict__[name] = call
        return call


@contextlib.contextmanager
def lazy_dma(ignore_waits=False):
    """inside: a DMA-to-LDS instruction writes LDS only when an `s_waitcnt vmcnt` of its wave forces it (the latest the
    hardware may), so a wait that is missing in front of a barrier shows as a wrong result.  ignore_waits=True drops every
    wait on top (self-test: such a kernel must FAIL)"""
    lib = load_emu()
    lib.emu_set_lazy_dma(2 if ignore_waits else 1)
    try:
        yield
    finally:
        lib.emu_set_lazy_dma(0)


@contextlib.contextmanager
def emulated_ops():
    """inside the context `ops.*` launches run in the emulator on CPU tensors"""
    from rsprompter_amd import _lib, ops
    lib = load_emu()
    import torch
    saved = (_lib._lib, ops._stream, ops._chk_f32, ops.require_device, ops._is_device)
    saved_sync = torch.cuda.synchronize
    _lib._lib = lib
    torch.cuda.synchronize = lambda *a, **k: None          # launches complete before they return

    def chk(t, name):
        import torch
        if t.dtype != torch.float32:
            raise ValueError(f'{name}: expected float32')
    ops._stream, ops._chk_f32, ops.require_device, ops._is_device = (lambda: 0), chk, (lambda dev: None), (lambda t: True)
    lib.emu_set_lazy_dma(1 if os.environ.get('RSP_WAVE_EMU_LAZY') == '1' else 0)     # audit mode: see lazy_dma()
    # round 6: LDS starts as 0xFF bytes (NaN) -- at every launch by default, at every BLOCK with RSP_WAVE_EMU_POISON_LDS=2 (audit:
    # an 8 MB memset per block doubles the suite's time): a read of LDS nobody wrote shows up instead of a plausible stale value
    lib.emu_set_poison_lds(int(os.environ.get('RSP_WAVE_EMU_POISON_LDS', '1')))
    try:
        yield ops
    finally:
        _lib._lib, ops._stream, ops._chk_f32, ops.require_device, ops._is_device = saved
        torch.cuda.synchronize = saved_sync
