#!/usr/bin/env python3
"""Start and end of consecutive kernels from a rocprofv3 --kernel-trace CSV, in us from the first one
shown (which launches overlap, and where the gaps are):
    python tools/ktimeline.py <dir or csv> [first record] [records]"""
import csv
import glob
import os
import sys

src = sys.argv[1]
if os.path.isdir(src):
    src = glob.glob(os.path.join(src, "**", "*_kernel_trace.csv"), recursive=True)[0]
first = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
count = int(sys.argv[3]) if len(sys.argv) > 3 else 16
rows = sorted(csv.DictReader(open(src)), key=lambda r: int(r["Start_Timestamp"]))
rows = rows[first:first + count]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    name = r["Kernel_Name"].replace("lmcma::(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    a, b = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    print("%8.2f .. %8.2f  (%6.2f)  queue %s  %s" % (a, b, b - a, r.get("Queue_Id", "?"), name[:60]))
