#!/usr/bin/env python3
"""Condense a rocprofv3 collection made by scripts/profile_bench.sh (gpurun_out/prof/<tag>_*) into
the small, tracked summaries under profiles/:
  profiles/<tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary (all kernels, name truncated)
  profiles/<tag>_pmc.json           per-kernel mean FETCH_SIZE / WRITE_SIZE (KiB as reported, separate passes)
  profiles/<tag>_bench.json         the bench line printed by the profiled run
  profiles/traffic.json             HBM bytes per launch of the seed-pass match kernel (read by bench.py)
usage: python scripts/summarize_prof.py <tag>
"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = os.path.join(ROOT, "gpurun_out", "prof")
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)

stats = glob.glob(os.path.join(src, tag + "_stats", "*", "*_kernel_stats.csv"))[0]
with open(stats) as f, open(os.path.join(dst, tag + "_kernel_stats.csv"), "w") as g:
    r = csv.reader(f)
    w = csv.writer(g)
    for row in r:
        row[0] = row[0][:160]
        w.writerow(row)

bench = json.load(open(os.path.join(src, tag + "_stats.json")))
json.dump(bench, open(os.path.join(dst, tag + "_bench.json"), "w"), indent=1)

pmc = {}
for sub, ctr in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    files = glob.glob(os.path.join(src, tag + "_" + sub, "*", "*_counter_collection.csv"))
    if not files:
        continue
    agg = collections.defaultdict(list)
    for row in csv.DictReader(open(files[0])):
        if row["Counter_Name"] == ctr:
            agg[row["Kernel_Name"][:120]].append(float(row["Counter_Value"]))
    for k, v in agg.items():
        if "nvbio_amd" in k:
            if "fm_match_kernel<4, false, true" in k and ctr == "FETCH_SIZE":
                # bench.py also launches this kernel with NVBIO_FM_NO_KMER_TABLE (the reference's algorithm, outside
                # the timed region): those launches fetch ~4.6x more and are reported on their own
                lo = min(v)
                no_table = [x for x in v if x > 1.5 * lo]
                v = [x for x in v if x <= 1.5 * lo]
                if no_table:
                    pmc.setdefault(k, {})["FETCH_SIZE_KiB_mean_no_table_launches"] = sum(no_table) / len(no_table)
            pmc.setdefault(k, {})[ctr + "_KiB_mean"] = sum(v) / len(v)
            pmc[k]["launches_" + sub] = len(v)
# issue / stall / L2 counters (means per launch), for the kernels of the hot path
for sub in ("sq", "tcc", "utcl", "ea"):
    files = glob.glob(os.path.join(src, tag + "_" + sub, "*", "*_counter_collection.csv"))
    if not files:
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(files[0])):
        agg[row["Kernel_Name"][:120]][row["Counter_Name"]].append(float(row["Counter_Value"]))
    for k, d in agg.items():
        if "nvbio_amd" in k and any(x in k for x in ("fm_match_kernel", "fm_seed_", "gotoh", "fm_filter_locate")):
            for c, v in d.items():
                pmc.setdefault(k, {})[c + "_mean"] = sum(v) / len(v)
json.dump(pmc, open(os.path.join(dst, tag + "_pmc.json"), "w"), indent=1, sort_keys=True)

# the seed-pass kernel of the timed region: the direct (match + locate fused) instantiation when the run used it
key = ([k for k in pmc if "fm_seed_both_kernel<4, false, true>" in k] or [k for k in pmc if "fm_seed_both_kernel<4, false, false>" in k] or [k for k in pmc if "fm_seed_pipe_kernel<4>" in k] or [k for k in pmc if "fm_seed_tiles_kernel<4, false>" in k]
       or [k for k in pmc if "fm_seed_diagonals_kernel<4>" in k] or [k for k in pmc if "fm_match_kernel<4, false, true, true>" in k]
       or [k for k in pmc if "fm_match_kernel<4, false, true" in k])
if key:
    m = pmc[key[0]]
    # FETCH_SIZE / WRITE_SIZE are reported in KiB.  WRITE_SIZE is exact here (90 M x 8 B of ranges per
    # launch = 703,125 KiB).  FETCH_SIZE is taken as reported: the kernel's reads are random 32-byte
    # records, each served as one 64-byte request, which is the unit the counter tallies (the x2
    # correction of MI355X_MICROARCH.md applies to wide 16 B/lane streams fetched as 128-byte requests).
    hbm = (m.get("FETCH_SIZE_KiB_mean", 0.0) + m.get("WRITE_SIZE_KiB_mean", 0.0)) * 1024.0
    cfg = bench["config"]
    json.dump({"tag": tag, "ref_len": cfg["ref_len"], "reads": cfg["reads_per_gpu"], "kmer": cfg["kmer_table"],
               "sa_int": cfg.get("sa_int", 16), "direct": bool(cfg.get("match_direct", False)), "fused": bool(cfg.get("fused_seed_pass", False)), "kernel": key[0][:80],
               "kernel_tag": ("fm_seed_both_kernel<4, false, true>" if "fm_seed_both_kernel<4, false, true>" in key[0] else
                              "fm_seed_both_kernel<4, false, false>" if "fm_seed_both_kernel<4, false, false>" in key[0] else
                              "fm_seed_pipe_kernel<4>" if "fm_seed_pipe_kernel<4>" in key[0] else key[0].split("(")[0].replace("void nvbio_amd::", "")),
               "match_hbm_bytes_per_launch": hbm, "fetch_KiB": m.get("FETCH_SIZE_KiB_mean"),
               "write_KiB": m.get("WRITE_SIZE_KiB_mean"),
               "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes; see scripts/summarize_prof.py"},
              open(os.path.join(dst, "traffic.json"), "w"), indent=1)
print("wrote", sorted(os.listdir(dst)))
