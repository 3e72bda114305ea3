"""In-tree build of librsp_hip.so (hipcc, gfx950 only).

`python -m rsprompter_amd.build` or `__graft_entry__.build()`.  The library is
written next to this file so that it travels with the repo snapshot to the GPU
box (it is git-ignored, not gpurun-ignored).
"""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "librsp_hip.so")
STAMP = os.path.join(HERE, ".librsp_hip.stamp")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
    "-ffp-contract=off",           # bit-stable box/IoU arithmetic (NMS parity with the CPU reference)
    # round 6: hipcc's SLP vectoriser pairs scalar fp32 work into v_pk_*_f32 whose LOW result takes the HIGH
    # register of a source pair (op_sel) -- the instruction form that dropped addends in sam_upscale2_kernel whenever two waves
    # shared a SIMD.  Off for the library; `// hipcc-flags: -fslp-vectorize` re-enables it per file where it was measured to pay
    # and the form does not appear (tests/test_isa_guard_cpu.py scans the device assembly of every kernel for it)
    "-fno-slp-vectorize",
    "-Wno-unused-result",
]
# development build (tools/gemm_s2_exp.py time|trace): also instantiates the GEMM's ablation variants, which compute
# wrong results on purpose.  Part of the digest, so a product process never loads a development library.
if os.environ.get("RSP_DEV_BUILD") == "1":
    FLAGS.append("-DRSP_S2_ABLATIONS")


def file_flags(src):
    """Extra per-file flags: a `// hipcc-flags: ...` line among the first lines of the source."""
    with open(src) as f:
        for _ in range(5):
            line = f.readline()
            if line.startswith("// hipcc-flags:"):
                return line.split(":", 1)[1].split()
    return []


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _digest():
    h = hashlib.sha256()
    for p in sources() + [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith(".h")] + [
            os.path.join(HERE, "..", "include", "rsp_hip.h")]:
        with open(p, "rb") as f:
            h.update(os.path.basename(p).encode())      # names, not absolute paths: the tree moves (gpurun snapshot)
            h.update(f.read())
    h.update(" ".join(FLAGS).encode())
    return h.hexdigest()


def _up_to_date(dig):
    if os.path.exists(LIB) and os.path.exists(STAMP):
        with open(STAMP) as f:
            return f.read().strip() == dig
    return False


def build(force=False, verbose=True):
    """Compile + link under an exclusive file lock: N ranks of `bench.py --gpus N` / torchrun that all find a stale
    library at import time must not run hipcc into the same object files or replace the .so while another rank is
    dlopen()ing it.  The first rank builds (objects in `build/` -- only the lock holder writes there --, the library
    linked to a temporary name and moved into place with os.replace, the stamp written last); the others block on the lock
    and then find the library up to date."""
    import fcntl
    dig = _digest()
    if not force and _up_to_date(dig):
        return LIB
    if not os.path.exists(HIPCC):
        raise RuntimeError(f"hipcc not found at {HIPCC}; cannot build librsp_hip.so")
    with open(os.path.join(HERE, ".librsp_hip.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and _up_to_date(dig):          # somebody else built it while we waited
                return LIB
            return _build_locked(dig, verbose)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def _build_locked(dig, verbose):
    objs = []
    procs = []
    bdir = os.path.join(HERE, "build")
    os.makedirs(bdir, exist_ok=True)
    for src in sources():
        obj = os.path.join(bdir, os.path.basename(src) + ".o")
        cmd = [HIPCC] + [f for f in FLAGS if f != "-shared"] + file_flags(src) + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd)))
        objs.append(obj)
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {src}")
    tmp_lib = LIB + f".tmp{os.getpid()}"
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp_lib] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    if os.path.exists(STAMP):
        os.remove(STAMP)                 # never a stamp that describes another library than the one on disk
    os.replace(tmp_lib, LIB)
    tmp_stamp = STAMP + f".tmp{os.getpid()}"
    with open(tmp_stamp, "w") as f:
        f.write(dig)
    os.replace(tmp_stamp, STAMP)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
