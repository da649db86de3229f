#!/usr/bin/env python
"""Static resource report of the PRODUCT library's kernels (no GPU needed): every csrc/*.hip is compiled for gfx950 with
-save-temps and the code-object metadata of each kernel is tabulated — VGPRs / AGPRs / SGPRs, spills, scratch, static LDS — together
with its count of MFMA, LDS-DMA (`buffer_load ... lds`) and `ds_read_b128` instructions.  Usage:
    python tools/isa_report.py [--tools] > profiles/rNN_isa_resources.txt"""
import argparse
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "pantomatrix_amd", "csrc")
sys.path.insert(0, CSRC)
import build  # noqa: E402


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    return [re.sub(r"\(anonymous namespace\)::|emage_dev::", "", n).replace("void ", "") for n in out]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tools", action="store_true", help="the -DEMAGE_TOOLS build (every swept tile configuration)")
    args = ap.parse_args()
    extra = ["-DEMAGE_TOOLS"] if args.tools else []
    rows = []
    with tempfile.TemporaryDirectory() as tmp:
        for src in build.SOURCES:
            subprocess.run([build._hipcc(), *build.FLAGS, *extra, "-save-temps", "-c", os.path.join(CSRC, src), "-o", os.path.join(tmp, src + ".o")],
                           cwd=tmp, check=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
            asm = [f for f in os.listdir(tmp) if f.startswith(src.replace(".hip", "") + "-hip-amdgcn") and f.endswith(".s")]
            if not asm:
                continue
            text = open(os.path.join(tmp, asm[0])).read()
            # instruction counts per kernel body
            counts = {}
            for m in re.finditer(r"^(_Z\w+):.*?\n(.*?)s_endpgm", text, re.S | re.M):
                body = m.group(2)
                counts[m.group(1)] = (len(re.findall(r"\bv_mfma_", body)), len(re.findall(r"buffer_load_dword\w* .* lds", body)), len(re.findall(r"\bds_read_b128\b", body)))
            for m in re.finditer(r"- \.agpr_count:\s+(\d+).*?\.group_segment_fixed_size:\s+(\d+).*?\.name:\s+(\S+).*?\.private_segment_fixed_size:\s+(\d+).*?\.sgpr_count:\s+(\d+).*?"
                                 r"\.sgpr_spill_count:\s+(\d+).*?\.vgpr_count:\s+(\d+).*?\.vgpr_spill_count:\s+(\d+)", text, re.S):
                agpr, lds, name, scratch, sgpr, sspill, vgpr, vspill = m.groups()
                c = counts.get(name, (0, 0, 0))
                rows.append((src, name, int(vgpr), int(agpr), int(sgpr), int(vspill), int(sspill), int(scratch), int(lds), *c))
    names = demangle([r[1] for r in rows])
    print(f"# {'tools' if args.tools else 'product'} library, hipcc {' '.join(build.FLAGS + extra)}; {len(rows)} kernels; vgpr = arch VGPRs + AGPRs as allocated")
    print(f"{'file':14s} {'vgpr':>4s} {'agpr':>4s} {'sgpr':>4s} {'vsp':>3s} {'ssp':>3s} {'scr':>4s} {'lds(B)':>6s} {'mfma':>5s} {'dma':>4s} {'dsr128':>6s}  kernel")
    for r, n in sorted(zip(rows, names), key=lambda t: (t[0][0], t[1])):
        print(f"{r[0]:14s} {r[2]:4d} {r[3]:4d} {r[4]:4d} {r[5]:3d} {r[6]:3d} {r[7]:4d} {r[8]:6d} {r[9]:5d} {r[10]:4d} {r[11]:6d}  {n[:150]}")
    print(f"# kernels with spills or scratch: {sum(1 for r in rows if r[5] or r[6] or r[7])}")


if __name__ == "__main__":
    main()
