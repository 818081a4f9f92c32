"""Prints name, calls and average microseconds from a rocprofv3 kernel_stats.csv (names contain commas)."""
import csv, sys
for row in csv.DictReader(open(sys.argv[1])):
    name = row["Name"].replace("(anonymous namespace)::", "").replace("lmcma::", "")
    name = name.split("(")[0] if not name.startswith("void") else name[5:].split("(")[0]
    print("  %-46s %7s calls  %8.2f us" % (name[:46], row["Calls"], float(row["AverageNs"]) / 1e3))
