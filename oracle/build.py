"""Build the oracle's C restatement (gcc) -> oracle/_build/liboracle_ops.so.  Tests only."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
OUT_DIR = os.path.join(HERE, '_build')
LIB = os.path.join(OUT_DIR, 'liboracle_ops.so')
SRC = os.path.join(HERE, 'mmcv_ops.c')


def build(force=False):
    os.makedirs(OUT_DIR, exist_ok=True)
    if not force and os.path.exists(LIB) and os.path.getmtime(LIB) >= os.path.getmtime(SRC):
        return LIB
    subprocess.check_call(['gcc', '-O2', '-fPIC', '-shared', '-ffp-contract=off', '-fno-fast-math',
                           '-o', LIB, SRC, '-lm'])
    return LIB


if __name__ == '__main__':
    print(build(force=True))
