#!/usr/bin/env python3
"""Per-kernel means of the counters in rocprofv3 --pmc CSV outputs: python tools/pmc_summary.py <dir> ..."""
import collections, csv, glob, sys
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*_counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].replace("lmcma::(anonymous namespace)::", "").replace("void ", "").split("(")[0][:40]
            a = agg[k][r["Counter_Name"]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
        for k in sorted(agg):
            print(k)
            for c in sorted(agg[k]):
                n, v = agg[k][c]
                print("   %-28s %14.1f per launch (%d launches)" % (c, v / n, n))
