"""Build libocc_amd.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m occnet_amd.build [--force]

Every csrc/*.hip is compiled to its own object (in parallel, cached by content digest under lib/obj/), then linked:
editing one kernel recompiles one file.  The library lands in occnet_amd/lib/ (git-ignored, shipped to the GPU box
by gpurun).  The stamp is a digest of the sources' RELATIVE names and contents, so it is valid wherever the tree lies.
"""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(LIBDIR, "obj")
LIB = os.path.join(LIBDIR, "libocc_amd.so")
STAMP = os.path.join(LIBDIR, "libocc_amd.stamp")
HEADER = os.path.join(os.path.dirname(HERE), "include", "occnet_amd.h")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-command-line-argument"]


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _headers():
    return [HEADER] + sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h"))


def _file_digest(paths):
    h = hashlib.sha256()
    h.update(" ".join(FLAGS).encode())
    for p in paths:
        h.update(os.path.basename(p).encode())
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def _digest():
    return _file_digest(sources() + _headers())


def source_digest(*names):
    """Digest of the named csrc files + the csrc headers (common.h): keys a measurement (e.g.
    profiles/sca_gather_traffic.json) to the kernel source it was taken on."""
    hdrs = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h"))
    return _file_digest([os.path.join(CSRC, n) for n in names] + hdrs)[:16]


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _compile(src, headers, verbose):
    dig = _file_digest([src] + headers)
    base = os.path.splitext(os.path.basename(src))[0]
    obj = os.path.join(OBJDIR, f"{base}.o")
    tag = os.path.join(OBJDIR, f"{base}.digest")
    if os.path.exists(obj) and os.path.exists(tag):
        with open(tag) as f:
            if f.read().strip() == dig:
                return obj, False
    cmd = [hipcc()] + FLAGS + ["-c", "-o", obj, src]
    if verbose:
        print(" ".join(cmd), flush=True)
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        sys.stderr.write(res.stdout + res.stderr)
        raise RuntimeError(f"hipcc failed on {src}")
    with open(tag, "w") as f:
        f.write(dig)
    return obj, True


def build(force=False, verbose=False):
    os.makedirs(OBJDIR, exist_ok=True)
    dig = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(STAMP):
        with open(STAMP) as f:
            if f.read().strip() == dig:
                return LIB
    if force:
        for f in os.listdir(OBJDIR):
            os.remove(os.path.join(OBJDIR, f))
    headers = _headers()
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        objs = [o for o, _ in ex.map(lambda s: _compile(s, headers, verbose), sources())]
    cmd = [hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), flush=True)
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        sys.stderr.write(res.stdout + res.stderr)
        raise RuntimeError("hipcc failed linking libocc_amd.so")
    with open(STAMP, "w") as f:
        f.write(dig)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
