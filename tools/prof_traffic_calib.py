#!/usr/bin/env python
"""Launches with KNOWN byte counts for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on this code's own access patterns, and the
emage_gemm (EMAGE_H2) launches whose memory-side traffic the round-4 verdict asked to explain (run on the MI355X, tools library):

    rocprofv3 --kernel-trace --pmc FETCH_SIZE -d out -o pmc --output-format csv -- python tools/prof_traffic_calib.py --plan out/plan.json
    rocprofv3 --kernel-trace --pmc WRITE_SIZE ...                                  (separate pass: the two do not fit one)
    python tools/prof_traffic_calib.py --summarize out_fetch/..counter_collection.csv out_write/..counter_collection.csv plan.json

Every case is launched REPS times back to back; `--plan` writes the launch order (case, kernel-name substring, expected bytes) so that the
summary can attribute the per-dispatch counter rows by order.  Expected bytes come in two flavours: `algo` = every operand once, and
`model` = what eight non-coherent per-XCD L2s must pull over the fabric with the kernel's tile order (an XCD walks a contiguous run of
tiles, tile_n fastest: it touches its own slice of A's rows once and ALL of W; FETCH_SIZE counts fabric requests, Infinity-Cache hits
included — MI355X_MICROARCH.md, HBM section)."""
import argparse
import csv
import json
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REPS = 3


def cases():
    """(name, kernel substring, spec) — spec drives both the launch and the byte model."""
    c = []
    # streaming calibrators on our own kernels (16 B per lane): LayerNorm reads M x C x 4 and writes M x C x 4 (H2 image)
    c.append(("layernorm 65536x768", "layernorm_kernel", dict(kind="ln", m=65536, c=768)))
    c.append(("layernorm 4096x768", "layernorm_kernel", dict(kind="ln", m=4096, c=768)))
    # (i) one N-tile column: A and W are each read exactly once by the launch (W once per XCD)
    c.append(("gemm M=65536 N=192 K=768 (one tile column)", "gemm_h2_kernel", dict(kind="gemm", m=65536, n=192, k=768, cfg=100)))
    c.append(("gemm M=4096 N=192 K=768 (one tile column)", "gemm_h2_kernel", dict(kind="gemm", m=4096, n=192, k=768, cfg=100)))
    # (ii) out_proj with and without its residual, shipped tile (64 x 64, three blocks per CU)
    c.append(("out_proj +res(h2)", "gemm_h2_kernel", dict(kind="gemm", m=4096, n=768, k=768, res="h2")))
    c.append(("out_proj no res", "gemm_h2_kernel", dict(kind="gemm", m=4096, n=768, k=768)))
    # (iii) the same without the XCD remap (dispatch order = tile order: consecutive tiles land on different XCDs), and M-first runs
    c.append(("out_proj +res(h2), no XCD remap", "gemm_h2_kernel", dict(kind="gemm", m=4096, n=768, k=768, res="h2", dbg=32)))
    c.append(("out_proj +res(h2), M-first runs", "gemm_h2_kernel", dict(kind="gemm", m=4096, n=768, k=768, res="h2", dbg=64)))
    # the wide launches of the step
    c.append(("ffn1 768->1536", "gemm_h2_kernel", dict(kind="gemm", m=4096, n=1536, k=768, slope=0.0)))
    c.append(("ffn2 1536->768 +res(h2)", "gemm_h2_kernel", dict(kind="gemm", m=4096, n=768, k=1536, res="h2")))
    c.append(("qkv 768->2304 +vt", "gemm_h2_kernel", dict(kind="gemm", m=4096, n=2304, k=768, vt=1536)))
    c.append(("kv_all 768->12288 +vt", "gemm_h2_kernel", dict(kind="gemm", m=4096, n=12288, k=768, vt=6144)))
    c.append(("kv_all 768->12288 +vt, M-first runs", "gemm_h2_kernel", dict(kind="gemm", m=4096, n=12288, k=768, vt=6144, dbg=64)))
    c.append(("kv_all 768->12288 +vt, no XCD remap", "gemm_h2_kernel", dict(kind="gemm", m=4096, n=12288, k=768, vt=6144, dbg=32)))
    c.append(("head 768->256", "gemm_h2_kernel", dict(kind="gemm", m=4096, n=256, k=768)))
    return c


TILES = {100: (64, 192), 120: (64, 64), 170: (128, 192), 119: (128, 256), 113: (128, 128)}


def product_config(m, n):
    """gemm_h2.hip: h2_config_for, for the plain shapes used here."""
    if n >= 1024 and m >= 1024:
        if ((m + 127) // 128) * ((n + 191) // 192) >= 1024 and n % 192 == 0:
            return 170
        if ((m + 127) // 128) * ((n + 255) // 256) >= 512 and n % 256 == 0:
            return 119
        if n % 192 == 0:
            return 100
        return 113
    return 120


def byte_model(sp):
    """-> dict(read_algo, read_model, write) in bytes for one launch of a case."""
    if sp["kind"] == "ln":
        b = sp["m"] * sp["c"] * 4
        return dict(read_algo=b, read_model=b, write=b)
    m, n, k = sp["m"], sp["n"], sp["k"]
    a_b, w_b = m * k * 4, n * k * 4
    res_b = m * n * 4 if sp.get("res") else 0
    out_b = m * n * 4                                   # H2 image / fp32 V^T: 4 bytes per element either way
    bm, bn = TILES[sp.get("cfg") or product_config(m, n)]
    tm, tn = math.ceil(m / bm), math.ceil(n / bn)
    nblk = tm * tn
    dbg = sp.get("dbg", 0)
    # which (tile_m, tile_n) each XCD touches
    per_xcd = [set() for _ in range(8)]
    for b in range(nblk):
        xcd = b % 8
        if dbg & 32:
            bid = b
        else:
            q, r, idx = nblk // 8, nblk % 8, b // 8
            bid = (xcd * (q + 1) if xcd < r else r * (q + 1) + (xcd - r) * q) + idx
        t = (bid % tm, bid // tm) if dbg & 64 else (bid // tn, bid % tn)
        per_xcd[xcd].add(t)
    read = 0
    for s in per_xcd:
        rows = {t[0] for t in s}
        cols = {t[1] for t in s}
        read += len(rows) * bm * k * 4 + len(cols) * bn * k * 4          # every distinct A row-panel and W panel once per XCD (L2 permitting)
    return dict(read_algo=a_b + w_b + res_b, read_model=min(read, nblk * (bm + bn) * k * 4) + res_b, write=out_b)


def run(plan_path):
    import torch
    from pantomatrix_amd import _lib, ops
    from pantomatrix_amd._lib import H2
    lib = _lib.use_tools(True)
    dev = "cuda"
    g = torch.Generator().manual_seed(0)
    plan = []
    for name, kern, sp in cases():
        if sp["kind"] == "ln":
            x = ops.h2_pack(torch.randn(sp["m"], sp["c"], generator=g).to(dev))
            gamma, beta = torch.ones(sp["c"], device=dev), torch.zeros(sp["c"], device=dev)
            y = torch.empty(sp["m"], sp["c"], device=dev)
            torch.cuda.synchronize()
            for _ in range(REPS):
                ops.layernorm(H2, x, gamma, beta, 1e-5, y=y)
        else:
            m, n, k = sp["m"], sp["n"], sp["k"]
            a = ops.h2_pack(torch.randn(m, k, generator=g).to(dev))
            w, ws = ops.split_f16_weights_h2((torch.randn(n, k, generator=g) / k ** 0.5).to(dev))
            bias = torch.zeros(n, device=dev)
            slope = torch.full((n,), float(sp["slope"]), device=dev) if "slope" in sp else None
            vt0 = sp.get("vt")
            ncol = vt0 or n
            res = ops.h2_pack(torch.randn(m, n, generator=g).to(dev)) if sp.get("res") else None
            out = torch.empty(m, ncol, device=dev)
            out_t = torch.empty(m // 64, n - vt0, 64, device=dev) if vt0 else None
            lib.emage_set_tuning(4, sp.get("cfg", -1))
            lib.emage_set_tuning(1, sp.get("dbg", 0))
            torch.cuda.synchronize()
            for _ in range(REPS):
                ops.gemm(H2, a, w, bias, slope, res, out, None, out_t, n=n, cp=k, t_col0=vt0 or 0, t_rows=64 if vt0 else 0, lin=64, lout=64, m=m,
                         w_scale=ws, res_h2=bool(res is not None))
            lib.emage_set_tuning(4, -1)
            lib.emage_set_tuning(1, 0)
        torch.cuda.synchronize()
        plan.append(dict(case=name, kernel=kern, reps=REPS, **byte_model(sp)))
    if plan_path:
        json.dump(plan, open(plan_path, "w"), indent=1)


def summarize(fetch_csv, write_csv, plan_path, out_path):
    plan = json.load(open(plan_path))

    def per_case(path, ctr):
        by_id = {}
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] == ctr:              # one row per dispatch (several if the tool splits by XCC: summed)
                e = by_id.setdefault(int(r["Dispatch_Id"]), [r["Kernel_Name"], 0.0])
                e[1] += float(r["Counter_Value"])
        seqs = {}
        for _id in sorted(by_id):
            kname, v = by_id[_id]
            for key in {p["kernel"] for p in plan}:
                if key in kname:
                    seqs.setdefault(key, []).append(v)
        res, pos = [], {}
        for p in plan:
            i = pos.get(p["kernel"], 0)
            vals = seqs.get(p["kernel"], [])[i:i + p["reps"]]
            pos[p["kernel"]] = i + p["reps"]
            res.append(sum(vals[1:]) / max(1, len(vals) - 1) if len(vals) > 1 else (vals[0] if vals else float("nan")))    # first launch: cold L2 / Infinity Cache
        return res

    fetch, write = per_case(fetch_csv, "FETCH_SIZE"), per_case(write_csv, "WRITE_SIZE")
    out = []
    for p, f, w in zip(plan, fetch, write):
        fb, wb = f * 1024.0, w * 1024.0                   # the counters are in KB
        out.append(dict(case=p["case"], read_algo_mb=p["read_algo"] / 1e6, read_model_mb=p["read_model"] / 1e6, write_algo_mb=p["write"] / 1e6,
                        fetch_size_mb=fb / 1e6, write_size_mb=wb / 1e6,
                        fetch_x2_over_algo=2 * fb / p["read_algo"], fetch_x2_over_model=2 * fb / p["read_model"], write_over_algo=wb / p["write"]))
    json.dump(out, open(out_path, "w"), indent=1)
    print(f"{'case':46s} read algo / model MB | 2xFETCH MB (/algo, /model) | write MB | WRITE_SIZE MB (/algo)")
    for o in out:
        print(f"{o['case']:46s} {o['read_algo_mb']:8.1f} {o['read_model_mb']:8.1f} | {2 * o['fetch_size_mb']:8.1f} ({o['fetch_x2_over_algo']:.2f}, {o['fetch_x2_over_model']:.2f}) | "
              f"{o['write_algo_mb']:7.1f} | {o['write_size_mb']:7.1f} ({o['write_over_algo']:.2f})")


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--plan", default="")
    ap.add_argument("--summarize", nargs=3, metavar=("FETCH_CSV", "WRITE_CSV", "PLAN"))
    ap.add_argument("--out", default="traffic_calibration.json")
    ap.add_argument("--model-only", action="store_true", help="print the byte model of every case (no GPU)")
    args = ap.parse_args()
    if args.model_only:
        for name, _k, sp in cases():
            print(f"{name:46s}", {k: round(v / 1e6, 1) for k, v in byte_model(sp).items()})
    elif args.summarize:
        summarize(*args.summarize, args.out)
    else:
        run(args.plan)
