"""Round 6, DESIGN 9.1: which instruction class of the round-5 `sam_upscale2_kernel` must be spaced out for the rare wrong
quarter-wave sums to disappear?  The failing build (round 5's source: SLP-packed sub-pixel sums, two blocks per CU) is compiled
to device assembly ONCE; every variant is that assembly with `s_nop`s inserted behind one class of instructions inside the
kernel -- nothing else moves, register allocation and schedule stay the failing ones -- assembled, bundled and linked into a
variant library next to the product one:

    python tools/probes/up2_isa_bisect.py            -> rsprompter_amd/variants/librsp_hip_<name>.so
    (GPU box)  for v in ...; do cp rsprompter_amd/variants/librsp_hip_$v.so rsprompter_amd/librsp_hip.so;
                              python tools/multimask_loop.py --iters 3000; done          (tools/r6_isa_bisect.sh)

Variants: base (unpatched), pk (behind every v_pk_*), acc (behind every v_accvgpr_read), trans (s_nop 3 behind every v_exp),
mov (behind every v_mov_b32), mfma (s_nop 7 behind every v_mfma), valu (behind EVERY v_* except MFMA: the blanket case),
lds (behind every ds_*), vmem (behind every global_load).  Test infrastructure; the product library is not touched."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, 'rsprompter_amd', 'csrc')
OUT = os.path.join(ROOT, 'rsprompter_amd', 'variants')
LLVM = '/opt/rocm/lib/llvm/bin'
HIPCC = '/opt/rocm/bin/hipcc'
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=off', '-Wno-unused-result']
KERNEL = '_ZN12_GLOBAL__N_119sam_upscale2_kernelENS_4Up2PE'
R5 = '92ce171'                                   # the commit whose upscale.hip holds the failing form

VARIANTS = {
    'base': [],
    'pk': [(r'^\s+v_pk_', 's_nop 0')],
    'acc': [(r'^\s+v_accvgpr_read', 's_nop 0')],
    'trans': [(r'^\s+v_exp_f32', 's_nop 3')],
    'mov': [(r'^\s+v_mov_b32', 's_nop 0')],
    'mfma': [(r'^\s+v_mfma', 's_nop 7')],
    'valu': [(r'^\s+v_(?!mfma)', 's_nop 0')],
    'lds': [(r'^\s+ds_', 's_nop 0')],
    'vmem': [(r'^\s+global_load', 's_nop 0')],
    # hypothesis variants, combined with a slowed VALU stream as `pk+vm0`: every global load waited for at once (no load in
    # flight behind any later instruction), every LDS read waited for at once
    'vm0': [(r'^\s+global_load', 's_waitcnt vmcnt(0)')],
    'lg0': [(r'^\s+ds_', 's_waitcnt lgkmcnt(0)')],
    # IN FRONT of the ds_bpermute_b32 that exchange the sub-pixel sums between the half waves (`bpN:0-1` = the first one only)
    # the wave gives up its issue slot only right behind / right in front of the eight  v_pk_mul_f32 ... op_sel:[0,1]
    'sel': [(r'^\s+v_pk_mul_f32 .*op_sel:\[0,1\]', 's_nop 0')], 'presel': [(r'^\s+v_pk_mul_f32 .*op_sel:\[0,1\]', '<s_nop 0')],
    'bp0': [(r'^\s+ds_bpermute', '<s_nop 0')], 'bp1': [(r'^\s+ds_bpermute', '<s_nop 1')],
    'bp3': [(r'^\s+ds_bpermute', '<s_nop 3')], 'bp7': [(r'^\s+ds_bpermute', '<s_nop 7')],
}


def run(cmd, **kw):
    subprocess.check_call(cmd, **kw)


def patch(asm, rules, lo=0, hi=1 << 30):
    """insert the rule's s_nop behind every matching instruction between the kernel's label and its .amdhsa_kernel block
    (`name:lo-hi`: only behind the matches number lo .. hi - 1 in program order)"""
    out, inside, n, k = [], False, 0, 0
    for line in asm.splitlines():
        if line.startswith(KERNEL + ':'):
            inside = True
        elif inside and '.amdhsa_kernel' in line:
            inside = False
        before = None
        if inside:
            for pat, nop in rules:
                if re.match(pat, line):
                    if lo <= k < hi:
                        n += 1
                        if nop.startswith('<'):              # '<...': in front of the instruction
                            out.append('\t' + nop[1:])
                        else:
                            before = '\t' + nop
                    k += 1
                    break
        out.append(line)
        if before:
            out.append(before)
    return '\n'.join(out) + '\n', n


def main():
    only = sys.argv[1:] or list(VARIANTS)
    work = os.path.join(OUT, 'work')
    os.makedirs(os.path.join(work, 'rsprompter_amd', 'csrc'), exist_ok=True)
    os.makedirs(os.path.join(work, 'include'), exist_ok=True)
    src = os.path.join(work, 'rsprompter_amd', 'csrc', 'upscale.hip')
    with open(src, 'w') as f:
        f.write(subprocess.check_output(['git', '-C', ROOT, 'show', f'{R5}:rsprompter_amd/csrc/upscale.hip'], text=True))
    for h, d in (('rsp_common.h', os.path.join('rsprompter_amd', 'csrc')), ('rsp_hip.h', 'include')):
        with open(os.path.join(ROOT, d, h)) as fi, open(os.path.join(work, d, h), 'w') as fo:
            fo.write(fi.read())
    dev_s = os.path.join(work, 'dev.s')
    run([HIPCC] + FLAGS + ['--cuda-device-only', '-S', src, '-o', dev_s])
    asm = open(dev_s).read()
    objs = [os.path.join(ROOT, 'rsprompter_amd', 'build', o) for o in sorted(os.listdir(os.path.join(ROOT, 'rsprompter_amd', 'build')))
            if o.endswith('.hip.o') and o != 'upscale.hip.o']
    for name in only:
        text, n = asm, 0
        for part in name.split('+'):                     # `a+b`: both rule sets; `a:lo-hi`: matches lo .. hi - 1 only
            if part == 'nosel':
                # every  v_pk_mul_f32 vD, vA, vB op_sel:[0,1]  (low = A.lo x B.HI, high = A.hi x B.hi) as two v_mul_f32 -- the
                # same two IEEE products, no packed instruction with a cross-selected source
                pat = re.compile(r'^(\s+)v_pk_mul_f32 v\[(\d+):(\d+)\], v\[(\d+):(\d+)\], v\[(\d+):(\d+)\] op_sel:\[0,1\]\s*$', re.M)
                text, k = pat.subn(lambda m: f'{m.group(1)}v_mul_f32_e32 v{m.group(2)}, v{m.group(4)}, v{m.group(7)}\n'
                                             f'{m.group(1)}v_mul_f32_e32 v{m.group(3)}, v{m.group(5)}, v{m.group(7)}', text)
                n += k
                continue
            if part == 'refresh':
                # in front of every  v_pk_mul_f32 vD, vA, vB op_sel:[0,1] : both registers of vB copied onto themselves -- the
                # hyper values were written by a global load long before; now their last writer is the VALU
                pat = re.compile(r'^(\s+)(v_pk_mul_f32 v\[\d+:\d+\], v\[\d+:\d+\], v\[(\d+):(\d+)\] op_sel:\[0,1\])\s*$', re.M)
                text, k = pat.subn(lambda m: f'{m.group(1)}v_mov_b32_e32 v{m.group(3)}, v{m.group(3)}\n'
                                             f'{m.group(1)}v_mov_b32_e32 v{m.group(4)}, v{m.group(4)}\n{m.group(1)}{m.group(2)}', text)
                n += k
                continue
            if part in ('nomfma', 'noexp', 'nopkfma'):
                # reduction of the reproducer (the kernel alone under the stress, tools/upscale2_loop.py: launches that differ from
                # the first): one ingredient replaced by constant writes to its destination registers -- the arithmetic becomes
                # meaningless but stays deterministic, addresses and control flow are untouched
                def regs(m_lo, m_hi, kind):
                    return [f'{kind}{i}' for i in range(int(m_lo), int(m_hi) + 1)]
                i0 = text.index(KERNEL + ':')
                i1 = text.index('.amdhsa_kernel ' + KERNEL)
                head, text, tail = text[:i0], text[i0:i1], text[i1:]         # this kernel's text only
                if part == 'nomfma':
                    # the first MFMA of a chain (C = 0) defines its 16 destination registers: constants there; the accumulating
                    # ones become a wait state
                    pat = re.compile(r'^(\s+)v_mfma_f32_32x32x16_f16 ([av])\[(\d+):(\d+)\], (.*)$', re.M)
                    text, k = pat.subn(lambda m: ('\n'.join(
                        f'{m.group(1)}' + (f'v_accvgpr_write_b32 {r}, 1.0' if m.group(2) == 'a' else f'v_mov_b32_e32 {r}, 1.0')
                        for r in regs(m.group(3), m.group(4), m.group(2))) if m.group(5).rstrip().endswith(', 0')
                        else f'{m.group(1)}s_nop 0'), text)
                elif part == 'noexp':
                    pat = re.compile(r'^(\s+)v_exp_f32_e\d+ (v\d+), .*$', re.M)
                    text, k = pat.subn(lambda m: f'{m.group(1)}v_mov_b32_e32 {m.group(2)}, 1.0', text)
                else:
                    pat = re.compile(r'^(\s+)v_pk_fma_f32 v\[(\d+):(\d+)\],.*$', re.M)
                    text, k = pat.subn(lambda m: f'{m.group(1)}v_mov_b32_e32 v{m.group(2)}, 0.5\n{m.group(1)}v_mov_b32_e32 v{m.group(3)}, 0.25', text)
                text = head + text + tail
                n += k
                continue
            if part == 'onecu':
                # 96 KB of static LDS in the kernel descriptor and the metadata: ONE block per CU, i.e. one wave per SIMD,
                # with the instruction stream untouched
                assert text.count('group_segment_fixed_size 32768') == 1 and text.count('group_segment_fixed_size: 32768') == 1
                text = text.replace('group_segment_fixed_size 32768', 'group_segment_fixed_size 98304')
                text = text.replace('group_segment_fixed_size: 32768', 'group_segment_fixed_size: 98304')
                continue
            base, _, rng = part.partition(':')
            lo, hi = (int(v) for v in rng.split('-')) if rng else (0, 1 << 30)
            text, k = patch(text, VARIANTS[base], lo, hi)
            n += k
        name = name.replace(':', '_').replace('+', '_')
        ps = os.path.join(work, f'dev_{name}.s')
        open(ps, 'w').write(text)
        dobj, dout, fb = (os.path.join(work, f'dev_{name}.{e}') for e in ('o', 'out', 'hipfb'))
        run([f'{LLVM}/clang', '-x', 'assembler', '-target', 'amdgcn-amd-amdhsa', '-mcpu=gfx950', '-c', ps, '-o', dobj])
        run([f'{LLVM}/lld', '-flavor', 'gnu', '-m', 'elf64_amdgpu', '--no-undefined', '-shared', '-o', dout, dobj])
        run([f'{LLVM}/clang-offload-bundler', '-type=o', '-bundle-align=4096',
             '-targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950', '-input=/dev/null', f'-input={dout}',
             f'-output={fb}'])
        hobj = os.path.join(work, f'upscale_{name}.o')
        run([HIPCC] + FLAGS + ['--cuda-host-only', '-Xclang', '-fcuda-include-gpubinary', '-Xclang', fb, '-c', src, '-o', hobj])
        lib = os.path.join(OUT, f'librsp_hip_{name}.so')
        run([HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', lib] + objs + [hobj])
        print(f'{name}: {n} s_nop lines inserted -> {os.path.relpath(lib, ROOT)}', flush=True)


if __name__ == '__main__':
    main()
