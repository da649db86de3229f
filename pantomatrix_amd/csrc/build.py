"""Build libemage_hip.so (gfx950) in-tree with hipcc.  No torch headers are involved: the library is a
plain C-ABI shared object (include/emage_hip.h) loaded by ctypes from pantomatrix_amd/_lib.py."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = ["gemm.hip", "gemm_h2.hip", "attention.hip", "vq.hip", "elementwise.hip", "motion.hip", "wavconv.hip", "convslab.hip", "lstm.hip", "lstmseq.hip", "train.hip", "version.hip"]
LIB = os.path.join(HERE, "libemage_hip.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++20", "-fPIC", "-ffp-contract=off", "-Wno-unused-result"]


def _hipcc():
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(HERE, s) for s in SOURCES] + [os.path.join(HERE, "common.h"), os.path.join(HERE, "gemm_tile.h"), os.path.join(HERE, "h2_tile.h"), os.path.join(HERE, "h2.h"), os.path.join(HERE, "attn_tile.h"), os.path.join(HERE, "ln_row.h"), os.path.join(HERE, "rot_math.h"),
                                                        os.path.join(HERE, "..", "..", "include", "emage_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not needs_build():
        return LIB
    objs = []
    procs = []
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    for s in SOURCES:
        o = os.path.join(HERE, "build", s.replace(".hip", ".o"))
        objs.append(o)
        procs.append((s, subprocess.Popen([_hipcc(), *FLAGS, "-c", os.path.join(HERE, s), "-o", o],
                                          stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
    for s, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {s}:\n{out}")
        if verbose and out.strip():
            print(out)
    r = subprocess.run([_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}")
    if verbose:
        print("built", LIB)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
