"""Is the gfx950 code of a kernel source the same as at a git revision?

    python tests/wave_emu/same_device_code.py [--rev HEAD] gemm_s2.hip relpos.hip ...

Compiles the revision's and the working tree's version of each source with the product flags to device assembly and
compares them instruction for instruction (comments, debug directives and the compilation-unit id symbol aside).  Used to
certify edits that must not change what the GPU runs -- the RSP_WAVE_LOCKSTEP() annotations for the lane-level emulator,
comment / host-code changes -- so that measurements and GPU test logs taken before the edit still describe the library."""
import argparse
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def device_asm(src, out):
    from rsprompter_amd import build as b
    flags = [f for f in b.FLAGS if f != '-shared']
    subprocess.check_call([b.HIPCC] + flags + b.file_flags(src) + ['--cuda-device-only', '-S', src, '-o', out],
                          stderr=subprocess.DEVNULL)
    t = open(out).read()
    t = re.sub(r';.*', '', t)
    t = re.sub(r'\.file.*|\.loc.*|\.ident.*', '', t)
    t = re.sub(r'__hip_cuid_[0-9a-f]+', '__hip_cuid', t)
    return [ln.rstrip() for ln in t.splitlines() if ln.strip()]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--rev', default='HEAD')
    ap.add_argument('sources', nargs='+')
    a = ap.parse_args()
    rc = 0
    with tempfile.TemporaryDirectory() as tmp:
        csrc = os.path.join(tmp, 'rsprompter_amd', 'csrc')
        os.makedirs(csrc)
        os.makedirs(os.path.join(tmp, 'include'))
        for rel in ['include/rsp_hip.h'] + ['rsprompter_amd/csrc/' + f for f in os.listdir(os.path.join(ROOT, 'rsprompter_amd', 'csrc'))
                                            if f.endswith('.h')]:
            open(os.path.join(tmp, rel), 'wb').write(subprocess.check_output(['git', '-C', ROOT, 'show', f'{a.rev}:{rel}']))
        for name in a.sources:
            rel = 'rsprompter_amd/csrc/' + name
            open(os.path.join(tmp, rel), 'wb').write(subprocess.check_output(['git', '-C', ROOT, 'show', f'{a.rev}:{rel}']))
            old = device_asm(os.path.join(tmp, rel), os.path.join(tmp, name + '.old.s'))
            new = device_asm(os.path.join(ROOT, rel), os.path.join(tmp, name + '.new.s'))
            same = old == new
            print(f'{name}: {len(new)} lines of device assembly, {"IDENTICAL to" if same else "DIFFERENT from"} {a.rev}')
            rc |= 0 if same else 1
    return rc


if __name__ == '__main__':
    sys.exit(main())
