"""TEST INFRASTRUCTURE: compile the kernel sources of rsprompter_amd/csrc UNCHANGED for the host against the lane-level
emulator (emu_hip.h) -> tests/wave_emu/_build/libemu_rsp.so with the same C entry points as librsp_hip.so.

The sources are not edited; a textual pre-pass over a copy does three things the host compiler needs:
  * `asm volatile(...)` statements: `s_waitcnt vmcnt(n)` becomes the emulator's wait (it drives the lazy-DMA mode), the
    empty optimisation fences become empty statements;
  * `__builtin_amdgcn_*` becomes `emu_amdgcn_*` (the host clang has no such builtins; emu_hip.h defines the functions);
  * `extern __shared__ T name[];` becomes a pointer to the emulator's dynamic-LDS buffer;
  * `__attribute__((amdgpu_waves_per_eu(..)))` (an occupancy hint) is dropped.
Nothing under rsprompter_amd/ imports this; only tests/test_wave_emu_cpu.py does."""
import hashlib
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, 'rsprompter_amd', 'csrc')
OUT = os.path.join(HERE, '_build')
CLANG = os.environ.get('EMU_CXX', '/opt/rocm/lib/llvm/bin/clang++')
DEFAULT_SOURCES = ['upscale.hip', 'norm.hip', 'gemm_s2.hip', 'gemm_dma.hip', 'attn_win.hip', 'attn_stream.hip', 'samattn.hip',
                   'det.hip', 'query.hip', 'misc.hip', 'elementwise.hip', 'rle.hip']


def strip_asm(src):
    """replace every `asm volatile( ... );` statement by `;` (balanced parentheses, string literals skipped)"""
    out, i = [], 0
    pat = re.compile(r'\basm\s+volatile\s*\(')
    while True:
        m = pat.search(src, i)
        if not m:
            out.append(src[i:])
            break
        out.append(src[i:m.start()])
        j, depth, in_str = m.end(), 1, False
        while depth:
            c = src[j]
            if in_str:
                if c == '\\':
                    j += 1
                elif c == '"':
                    in_str = False
            elif c == '"':
                in_str = True
            elif c == '(':
                depth += 1
            elif c == ')':
                depth -= 1
            j += 1
        text = src[m.end():j - 1]
        w = re.search(r'"s_waitcnt vmcnt\((%0|\d+)\)"', text)
        if w and w.group(1) == '%0':                     # asm volatile("s_waitcnt vmcnt(%0)" ::"n"(EXPR) : "memory")
            e = re.search(r'"n"\((.*)\)\s*:', text, re.S)
            out.append(f'emu_vmcnt_wait({e.group(1)})')
        elif w:
            out.append(f'emu_vmcnt_wait({w.group(1)})')
        else:
            out.append('((void)0)')
        i = j
    return ''.join(out)


def prepass(text):
    text = strip_asm(text)
    text = text.replace('__builtin_amdgcn_', 'emu_amdgcn_')
    text = text.replace('__hip_atomic_fetch_add(', 'emu_hip_atomic_fetch_add(')
    text = re.sub(r'__attribute__\(\(amdgpu_waves_per_eu\([^)]*\)\)\)', '', text)
    text = re.sub(r'extern\s+__shared__\s+(?:__attribute__\(\(aligned\(\d+\)\)\)\s+)?([A-Za-z_][\w ]*?)\s+(\w+)\[\];',
                  r'\1* \2 = reinterpret_cast<\1*>(emu::dyn_smem());', text)
    return text


def build(sources=None, verbose=False):
    sources = sources or DEFAULT_SOURCES
    os.makedirs(OUT, exist_ok=True)
    h = hashlib.sha256()
    for name in sorted(os.listdir(CSRC)) + ['emu_hip.h', 'build.py']:
        p = os.path.join(CSRC, name) if os.path.exists(os.path.join(CSRC, name)) else os.path.join(HERE, name)
        h.update(open(p, 'rb').read())
    h.update(' '.join(sources).encode())
    lib = os.path.join(OUT, 'libemu_rsp.so')
    stamp = os.path.join(OUT, 'digest.txt')
    if os.path.exists(lib) and os.path.exists(stamp) and open(stamp).read() == h.hexdigest():
        return lib
    objs = []
    flags = ['-std=c++17', '-O2', '-g', '-fPIC', '-ffp-contract=off', '-Wno-everything', '-fno-strict-aliasing',
             '-I', os.path.join(HERE, 'stub'), '-I', CSRC, '-I', os.path.join(ROOT, 'include')]
    for name in sources:
        text = prepass(open(os.path.join(CSRC, name)).read())
        cpp = os.path.join(OUT, name + '.emu.cpp')
        # the sources include "rsp_common.h" / "../../include/rsp_hip.h" relative to csrc: keep that working from _build
        text = text.replace('#include "rsp_common.h"', '#include "rsp_common.emu.h"')
        open(cpp, 'w').write(text)
        objs.append((cpp, os.path.join(OUT, name + '.o')))
    common = prepass(open(os.path.join(CSRC, 'rsp_common.h')).read())
    common = common.replace('#include "../../include/rsp_hip.h"', '#include "rsp_hip.h"')
    open(os.path.join(OUT, 'rsp_common.emu.h'), 'w').write(common)
    open(os.path.join(OUT, 'emu_impl.cpp'), 'w').write('#define EMU_IMPLEMENTATION 1\n#include <hip/hip_runtime.h>\n')
    objs.append((os.path.join(OUT, 'emu_impl.cpp'), os.path.join(OUT, 'emu_impl.o')))
    procs = []
    for cpp, obj in objs:
        cmd = [CLANG] + flags + ['-x', 'c++', '-c', cpp, '-o', obj]
        if verbose:
            print(' '.join(cmd))
        procs.append((cpp, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    failed = False
    for cpp, p in procs:
        out = p.communicate()[0].decode()
        if p.returncode:
            failed = True
            sys.stderr.write(f'--- {cpp}\n{out[:6000]}\n')
    if failed:
        raise RuntimeError('wave_emu build failed')
    # -Bsymbolic: the library's references to hipGetLastError, hipLaunch... bind to ITS OWN (inline) definitions even when
    # the real HIP runtime is already in the process (librsp_hip.so loads it with RTLD_GLOBAL in the same test session)
    cmd = [CLANG, '-shared', '-fPIC', '-Wl,-Bsymbolic', '-o', lib + '.tmp'] + [o for _, o in objs]
    subprocess.check_call(cmd)
    os.replace(lib + '.tmp', lib)
    open(stamp, 'w').write(h.hexdigest())
    return lib


if __name__ == '__main__':
    print(build(sys.argv[1:] or None, verbose=True))
