"""Build libemage_hip.so (gfx950) in-tree with hipcc.  No torch headers are involved: the library is a
plain C-ABI shared object (include/emage_hip.h) loaded by ctypes from pantomatrix_amd/_lib.py.

Two libraries come out of the same sources:
  libemage_hip.so        the PRODUCT: only the tile configurations the heuristics select, no tuning hooks, no mutable globals;
  libemage_hip_tools.so  the same plus -DEMAGE_TOOLS: every tile configuration, `emage_set_tuning`, the diagnostic ablation
                         branches and the phase tracer — used by tools/ and by the tests that walk every configuration."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = ["gemm.hip", "gemm_h2.hip", "gemm_h2_pp.hip", "attention.hip", "vq.hip", "elementwise.hip", "motion.hip", "wavconv.hip", "convslab.hip", "lstm.hip", "lstmseq.hip", "train.hip", "version.hip"]
HEADERS = ["common.h", "gemm_tile.h", "h2_tile.h", "h2_pp_tile.h", "h2.h", "attn_tile.h", "ln_row.h", "rot_math.h"]
LIB = os.path.join(HERE, "libemage_hip.so")
TOOLS_LIB = os.path.join(HERE, "libemage_hip_tools.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++20", "-fPIC", "-ffp-contract=off", "-Wno-unused-result"]


def _hipcc():
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def _deps():
    return [os.path.join(HERE, s) for s in SOURCES + HEADERS] + [os.path.join(HERE, "..", "..", "include", "emage_hip.h"), os.path.abspath(__file__)]


def needs_build(lib: str = None) -> bool:
    libs = [LIB, TOOLS_LIB] if lib is None else [lib]
    for l in libs:
        if not os.path.exists(l):
            return True
        t = os.path.getmtime(l)
        if any(os.path.getmtime(d) > t for d in _deps()):
            return True
    return False


def _includes(path, seen=None):
    """The in-tree headers a source pulls in (quoted #include lines, followed recursively)."""
    import re
    seen = set() if seen is None else seen
    try:
        text = open(path).read()
    except OSError:
        return seen
    for inc in re.findall(r'^\s*#\s*include\s+"([^"]+)"', text, re.M):
        q = os.path.normpath(os.path.join(os.path.dirname(path), inc))
        if q not in seen and os.path.exists(q):
            seen.add(q)
            _includes(q, seen)
    return seen


def _stale(src, obj):
    """An object is rebuilt when its source, one of the headers it includes, or this script is newer (or `force`)."""
    if not os.path.exists(obj):
        return True
    t = os.path.getmtime(obj)
    return any(os.path.getmtime(d) > t for d in [src, os.path.abspath(__file__), *_includes(src)])


def _build_one(lib, objdir, extra, verbose, force=False):
    objs, procs = [], []
    os.makedirs(objdir, exist_ok=True)
    for s in SOURCES:
        o = os.path.join(objdir, s.replace(".hip", ".o"))
        objs.append(o)
        if not force and not _stale(os.path.join(HERE, s), o):
            continue
        procs.append((s, subprocess.Popen([_hipcc(), *FLAGS, *extra, "-c", os.path.join(HERE, s), "-o", o],
                                          stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {s}:\n{out}")
        if verbose and out.strip():
            print(out)
    r = subprocess.run([_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", lib],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}")
    if verbose:
        print("built", lib)


def build(force: bool = False, verbose: bool = True, tools: bool = True, all_objects: bool = False) -> str:
    # `force` (a missing or out-of-date library): objects whose own dependencies did not change are kept — `python build.py --force --all` rebuilds everything
    if force or needs_build(LIB):
        _build_one(LIB, os.path.join(HERE, "build"), [], verbose, force=all_objects)
    if tools and (force or needs_build(TOOLS_LIB)):
        _build_one(TOOLS_LIB, os.path.join(HERE, "build_tools"), ["-DEMAGE_TOOLS"], verbose, force=all_objects)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, all_objects="--all" in sys.argv)
