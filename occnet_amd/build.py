"""Build libocc_amd.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m occnet_amd.build [--force]

The library lands in occnet_amd/lib/ (git-ignored, shipped to the GPU box by gpurun).
"""
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libocc_amd.so")
STAMP = os.path.join(LIBDIR, "libocc_amd.stamp")
HEADER = os.path.join(os.path.dirname(HERE), "include", "occnet_amd.h")


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _digest():
    h = hashlib.sha256()
    for p in sources() + [HEADER] + sorted(
            os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")):
        h.update(p.encode())
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def build(force=False, verbose=False):
    os.makedirs(LIBDIR, exist_ok=True)
    dig = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(STAMP):
        with open(STAMP) as f:
            if f.read().strip() == dig:
                return LIB
    cmd = [hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC",
           "-Wno-unused-command-line-argument", "-o", LIB] + sources()
    if verbose:
        print(" ".join(cmd), flush=True)
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        sys.stderr.write(res.stdout + res.stderr)
        raise RuntimeError("hipcc failed building libocc_amd.so")
    with open(STAMP, "w") as f:
        f.write(dig)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
