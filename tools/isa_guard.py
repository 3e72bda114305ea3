"""Static guard over the device code of rsprompter_amd/csrc (round 6).

The box-dependent wrong answer of round 5 was ONE instruction form: `v_pk_mul_f32 vD, vA, vB op_sel:[0,1]` -- a packed fp32
operation whose LOW result takes the HIGH register of a source pair -- losing its low product in lanes 48..63 when a second
wave of the same SIMD had an instruction issued next to it (tools/probes/up2_isa_bisect.py: the failing build with exactly
those eight instructions replaced by two v_mul_f32 each never fails, 0 of 3000 launches and 0 of 200 under the stress that
made every launch wrong; with one wave per SIMD the unchanged stream never fails either; tools/probes/pk_opsel_probe.hip
reproduces it with ten instructions: the form behind an MFMA of the same wave, the issue slot given up in between).  hipcc
emits the form from its SLP vectoriser and from `vector * scalar` on ext-vector types; nothing in the sources asks for it.

This script compiles every .hip with the library's flags to device assembly and lists, per kernel, the packed-fp32
instructions with a set `op_sel` bit and the occupancy the compiler reports.  A kernel that has such instructions AND can
share a SIMD (occupancy > 1) is a violation: exit code 1.   python tools/isa_guard.py [-v]"""
import concurrent.futures
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

PK_SEL = re.compile(r'^\s+v_pk_\w+_f32 .*\bop_sel:\[')
LABEL = re.compile(r'^([A-Za-z_]\w*):')
OCC = re.compile(r'^; Occupancy: (\d+)')


def device_asm(src, outdir):
    from rsprompter_amd import build
    out = os.path.join(outdir, os.path.basename(src) + '.s')
    flags = [f for f in build.FLAGS if f not in ('-shared',)] + build.file_flags(src)
    subprocess.check_call([build.HIPCC] + flags + ['--cuda-device-only', '-S', src, '-o', out],
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return out


def scan(asm_path):
    """{kernel: (instructions with a cross-selected packed-fp32 source, occupancy)} for the kernels of one file"""
    res, cur, hits = {}, None, []
    for line in open(asm_path):
        m = LABEL.match(line)
        if m and not line.startswith('.L'):
            cur, hits = m.group(1), []
            continue
        if cur is None:
            continue
        if PK_SEL.match(line):
            hits.append(line.strip())
        m = OCC.match(line)
        if m:
            res[cur] = (hits, int(m.group(1)))
            cur = None
    return res


def run(verbose=False):
    from rsprompter_amd import build
    viol, table = [], []
    with tempfile.TemporaryDirectory() as td:
        with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
            asms = list(ex.map(lambda s: device_asm(s, td), build.sources()))
        for a in asms:
            for k, (hits, occ) in sorted(scan(a).items()):
                if hits:
                    table.append((os.path.basename(a)[:-2], k, len(hits), occ))
                    if occ > 1:
                        viol.append((os.path.basename(a)[:-2], k, len(hits), occ, hits[:3]))
                elif verbose:
                    table.append((os.path.basename(a)[:-2], k, 0, occ))
    return table, viol


if __name__ == '__main__':
    table, viol = run('-v' in sys.argv)
    for f, k, n, occ in table:
        print(f'{f:18s} {k[:80]:80s} packed-fp32 with op_sel: {n:4d}   waves per SIMD: {occ}')
    for v in viol:
        print('VIOLATION:', v)
    print(f'{len(viol)} violation(s)')
    sys.exit(1 if viol else 0)
