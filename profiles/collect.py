#!/usr/bin/env python3
"""Condense rocprofv3 outputs (under gpurun_out/) into the small summaries kept here.

    python profiles/collect.py <tag> <kernel-trace dir> [<pmc FETCH_SIZE dir> <pmc WRITE_SIZE dir>]

Writes profiles/<tag>_kernel_stats.csv (the --kernel-trace --stats summary as is),
profiles/<tag>_hbm_traffic.csv (per kernel: launches, mean FETCH_SIZE and WRITE_SIZE
in KB per launch as the counters report them, and the corrected bytes per launch) and
profiles/traffic_cfg2.json (corrected bytes per launch per kernel class and their sum per
generation, read by bench.py).
Correction: FETCH_SIZE on gfx950 reports half the bytes of wide (16 B/lane) coalesced
streaming reads, which is how these kernels read everything of any size, so
bytes = 2 * FETCH_SIZE + WRITE_SIZE (both in KB of 1024 bytes).  Narrower reads (the
4-byte row gathers of the ranking kernel) are then counted at up to twice their size:
the figure is an upper bound on the traffic."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CLASS_OF = {"k_generate": "generate", "k_generate_pre": "generate", "k_generate_mfma": "generate",
            "k_generate_fused": "generate+update", "k_sample": "sample",
            "k_cost": "cost", "k_rank": "select", "k_rank_sorted": "select", "k_rank_merge": "select",
            "k_sort_tiles": "select", "k_merge_rank": "select", "k_psr_count": "select",
            "k_recombine_partial": "recombine", "k_recombine_final": "recombine", "k_reduce_rows": "recombine",
            "k_update": "update", "k_update_pre": "update", "k_update_post": "update"}


def short(name):
    return name.replace("lmcma::(anonymous namespace)::", "").replace("void ", "").split("(")[0].split("<")[0]


def pmc(directory):
    f = glob.glob(os.path.join(directory, "**", "*_counter_collection.csv"), recursive=True)[0]
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        agg[k][0] += 1
        agg[k][1] += float(r["Counter_Value"])
    return agg


def main():
    tag, trace = sys.argv[1], sys.argv[2]
    stats = glob.glob(os.path.join(trace, "**", "*_kernel_stats.csv"), recursive=True)[0]
    shutil.copy(stats, os.path.join(HERE, tag + "_kernel_stats.csv"))
    if len(sys.argv) >= 5:
        fe, wr = pmc(sys.argv[3]), pmc(sys.argv[4])
        per_class = collections.defaultdict(float)
        total, generations = 0.0, 0
        with open(os.path.join(HERE, tag + "_hbm_traffic.csv"), "w") as out:
            out.write("kernel,launches,FETCH_SIZE_KB_per_launch,WRITE_SIZE_KB_per_launch,corrected_bytes_per_launch\n")
            for k in sorted(fe):
                f_kb = fe[k][1] / fe[k][0]
                w_kb = wr[k][1] / wr[k][0] if k in wr else 0.0
                out.write("%s,%d,%.3f,%.3f,%.0f\n" % (k, fe[k][0], f_kb, w_kb, (2 * f_kb + w_kb) * 1024.0))
                if k in CLASS_OF:
                    per_class[CLASS_OF[k]] += (2 * f_kb + w_kb) * 1024.0
                    total += (2 * f_kb + w_kb) * 1024.0 * fe[k][0]
                    if CLASS_OF[k] == "select":
                        generations = max(generations, fe[k][0])   # every generation has one ranking launch
        # all launches of the run together, per generation (inside a call the update rides in the
        # next sampling launch, so the kernels are not launched equally often)
        per_class["generation"] = total / max(generations, 1)
        json.dump({k: round(v) for k, v in per_class.items()}, open(os.path.join(HERE, "traffic_cfg2.json"), "w"),
                  indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
