#!/usr/bin/env python3
"""Summaries of the rocprofv3 output collected by profiles/collect.sh (run HERE, after gpurun merged the raw
csv files back under gpurun_out/).

  python profiles/summarize.py stats gpurun_out/prof_stats  profiles/rNN   -> rNN_bench512_kernel_stats.csv,
                                                                              rNN_bench512_by_kernel_and_grid.txt
  python profiles/summarize.py pmc gpurun_out/prof_fetch gpurun_out/prof_write profiles/rNN [gpurun_out/prof_fetch_g gpurun_out/prof_write_g]
                                                                           -> rNN_pmc512_fetch_write.txt, traffic_rNN.json
"""
import csv
import glob
import json
import os
import re
import shutil
import sys
from collections import defaultdict


def short(name):
    m = re.search(r"namespace\)::([A-Za-z0-9_]+(<[^>]*>)?)", name)
    if m:
        return m.group(1)
    return name.split("(")[0].replace("void ", "")


def find(d, suffix):
    f = sorted(glob.glob(os.path.join(d, "*", "*" + suffix)), key=os.path.getmtime)
    if not f:
        raise SystemExit(f"no {suffix} under {d}")
    return f[-1]  # the most recent run


def stats(src, out):
    shutil.copy(find(src, "_kernel_stats.csv"), out + "_bench512_kernel_stats.csv")
    rows = list(csv.DictReader(open(find(src, "_kernel_trace.csv"))))
    agg = defaultdict(lambda: [0, 0.0])
    total = 0.0
    for r in rows:
        dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        key = (short(r["Kernel_Name"]), int(r["Grid_Size"]) if "Grid_Size" in r else int(r["Grid_Size_X"]))
        agg[key][0] += 1
        agg[key][1] += dur
        total += dur
    with open(out + "_bench512_by_kernel_and_grid.txt", "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-cpu\n")
        f.write("#   (laplace_3d 512^3 7-pt, GMRES(50)+BoomerAMG, 1x MI355X; setup kernels + 3 solves in the trace)\n")
        f.write("# per (kernel, grid size) = per AMG level: calls, mean us, total ms, share of GPU time\n")
        f.write("# spmv_stream_xc<0, 1, ., 256> = the level-0 operator of the GMRES loop (C-first ordering of level 0): the kernel bench.py reports as \"roofline\"\n")
        for (name, grid), (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            if us / total < 2e-4:
                continue
            f.write(f"{name:28s} grid {grid:<10d} calls {n:5d}  mean {us / n:10.1f} us  total {us / 1e3:9.1f} ms  {100 * us / total:5.1f}%\n")
    print("wrote", out + "_bench512_kernel_stats.csv", out + "_bench512_by_kernel_and_grid.txt")


def counter_means(src, counter):
    rows = list(csv.DictReader(open(find(src, "_counter_collection.csv"))))
    agg = defaultdict(lambda: [0, 0.0])
    for r in rows:
        if r["Counter_Name"] != counter:
            continue
        key = (short(r["Kernel_Name"]), int(r["Grid_Size"]))
        agg[key][0] += 1
        agg[key][1] += float(r["Counter_Value"])
    return {k: (n, v / n) for k, (n, v) in agg.items()}


def pmc(fetch_dir, write_dir, out, fetch_g=None, write_g=None):
    fe = counter_means(fetch_dir, "FETCH_SIZE")
    wr = counter_means(write_dir, "WRITE_SIZE")
    tag = os.path.basename(out)
    lines = []
    for key, (n, kb) in sorted(fe.items(), key=lambda kv: -kv[1][0] * kv[1][1]):
        f_gb = kb * 1024 / 1e9
        w_gb = wr.get(key, (0, 0.0))[1] * 1024 / 1e9
        if n * f_gb < 0.5:
            continue
        lines.append(f"{key[0]:28s} grid {key[1]:<10d} n={n:4d}  FETCH_raw {f_gb:8.3f} GB  x2 {2 * f_gb:8.3f} GB  WRITE {w_gb:7.3f} GB")
    with open(out + "_pmc512_fetch_write.txt", "w") as f:
        f.write("# separate passes: rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv ... and --pmc WRITE_SIZE ...\n")
        f.write("#   -- python3 bench.py --steps 1 --warmup 0 --no-cpu   (512^3 7-pt, 1x MI355X)\n")
        f.write("# raw counter means per launch (KB*1024 -> GB).  gfx950: FETCH_SIZE reports 1/2 of a wide coalesced read\n")
        f.write("# (MI355X_MICROARCH.md, HBM section): check it on axpy/scale kernels of known traffic below; WRITE_SIZE is exact.\n")
        f.write("\n".join(lines) + "\n")
    key = next(k for k in fe if k[0].startswith("spmv_stream_xc<0, 1") or k[0].startswith("spmv_stream<0, 1"))
    fetch_raw = fe[key][1] * 1024
    write = wr[key][1] * 1024
    js = {"512^3/7pt/1gpu": {
        "spmv_hbm_bytes_per_launch": 2 * fetch_raw + write,
        "fetch_size_raw_bytes": fetch_raw,
        "write_size_bytes": write,
        "kernel": key[0],
        "note": "FETCH_SIZE doubled per the gfx950 correction; separate --pmc passes; see profiles/%s_pmc512_fetch_write.txt. "
                "A kernel whose third template argument is true (spmv_stream_xc<EPI, TAG, VAL8, threads>) streams one-byte dictionary indices instead of 8-byte values (DESIGN.md "
                "section 5): its traffic can be BELOW the algorithmic 12 nnz + 20 N, which prices 8-byte values" % tag}}
    if fetch_g and write_g:
        # the same kernels with the value dictionary off (MI_HYPRE_VALUE_DICT=0): what a general operator moves
        feg = counter_means(fetch_g, "FETCH_SIZE")
        wrg = counter_means(write_g, "WRITE_SIZE")
        with open(out + "_pmc512_fetch_write.txt", "a") as f:
            f.write("\n# ---- value dictionary OFF (MI_HYPRE_VALUE_DICT=0): the general-operator stream of level 0\n")
            for k2, (n, kb) in sorted(feg.items(), key=lambda kv: -kv[1][0] * kv[1][1]):
                f_gb = kb * 1024 / 1e9
                w_gb = wrg.get(k2, (0, 0.0))[1] * 1024 / 1e9
                if n * f_gb < 0.5:
                    continue
                f.write(f"{k2[0]:28s} grid {k2[1]:<10d} n={n:4d}  FETCH_raw {f_gb:8.3f} GB  x2 {2 * f_gb:8.3f} GB  WRITE {w_gb:7.3f} GB\n")
        kg = next(k2 for k2 in feg if k2[0].startswith("spmv_stream_xc<0, 1") or k2[0].startswith("spmv_stream<0, 1"))
        js["512^3/7pt/1gpu"]["spmv_general_hbm_bytes_per_launch"] = 2 * feg[kg][1] * 1024 + wrg[kg][1] * 1024
        js["512^3/7pt/1gpu"]["general_kernel"] = kg[0]
    path = os.path.join(os.path.dirname(out), "traffic_%s.json" % tag)
    json.dump(js, open(path, "w"), indent=1)
    print("wrote", out + "_pmc512_fetch_write.txt", path)


if __name__ == "__main__":
    if sys.argv[1] == "stats":
        stats(sys.argv[2], sys.argv[3])
    else:
        pmc(sys.argv[2], sys.argv[3], sys.argv[4], *(sys.argv[5:7] if len(sys.argv) >= 7 else ()))
