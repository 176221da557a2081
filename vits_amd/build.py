"""Builds vits_amd/lib/libvits_amd.so from vits_amd/csrc with hipcc for gfx950.

The library is built in-tree (it is git-ignored but travels with the working tree to
the GPU box).  ``python -m vits_amd.build`` rebuilds; ``build(force=False)`` skips
translation units whose object file is newer than every source it includes.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(LIBDIR, "obj")
LIB = os.path.join(LIBDIR, "libvits_amd.so")
DIAG_LIB = os.path.join(LIBDIR, "libvits_amd_diag.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"
FLAGS = ["-O3", "-fPIC", "-std=c++17", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function"]


def csrc_hash() -> str:
    """Fingerprint of the kernel sources; stored measurements carry the one they were made with."""
    import hashlib
    h = hashlib.sha256()
    for f in sorted(os.listdir(CSRC)):
        if f.endswith((".hip", ".hpp", ".h")):
            h.update(f.encode())
            h.update(open(os.path.join(CSRC, f), "rb").read())
    return h.hexdigest()[:16]


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _headers_mtime():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hpp", ".h"))]
    hs.append(os.path.join(os.path.dirname(HERE), "include", "vits_amd.h"))
    return max(os.path.getmtime(h) for h in hs)


def _compile(src: str, force: bool, hdr_mtime: float, diag: bool = False) -> str:
    obj = os.path.join(OBJDIR + ("_diag" if diag else ""), src[:-4] + ".o")
    path = os.path.join(CSRC, src)
    if (not force and os.path.exists(obj)
            and os.path.getmtime(obj) > max(os.path.getmtime(path), hdr_mtime)):
        return obj
    extra = os.environ.get("VITS_AMD_HIPCC_FLAGS", "").split() if diag else []      # diagnostic experiments only
    cmd = [HIPCC, *FLAGS, *(["-DVK_DIAG"] if diag else []), *extra, "-c", path, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
    return obj


def build(force: bool = False, jobs: int | None = None, diag: bool = False) -> str:
    """The product library, or with ``diag`` the diagnostic build (``-DVK_DIAG``: environment
    switches for timing experiments and per-workgroup time stamps; only tools/ load it)."""
    lib = DIAG_LIB if diag else LIB
    os.makedirs(OBJDIR + ("_diag" if diag else ""), exist_ok=True)
    srcs = _sources()
    hdr_mtime = _headers_mtime()
    # a library newer than every source is up to date even where the object files did not travel
    newest = max([hdr_mtime] + [os.path.getmtime(os.path.join(CSRC, f)) for f in srcs])
    if not force and os.path.exists(lib) and os.path.getmtime(lib) > newest:
        return lib
    jobs = jobs or min(8, len(srcs))
    with ThreadPoolExecutor(max_workers=jobs) as ex:
        objs = list(ex.map(lambda s: _compile(s, force, hdr_mtime, diag), srcs))
    if force or not os.path.exists(lib) or any(os.path.getmtime(o) > os.path.getmtime(lib) for o in objs):
        cmd = [HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", lib, *objs]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, diag="--diag" in sys.argv))
