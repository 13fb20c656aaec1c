"""Prints a rocprofv3 kernel_stats.csv as a table: python tools/show_stats.py <csv> [calls_divisor]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
div = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total GPU time %.3f ms (%.3f ms per step)" % (tot / 1e6, tot / 1e6 / div))
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 40]:
    print("%-70s calls %5s  per-step ms %8.3f  avg us %9.1f  %5.1f%%" % (r["Name"][:70], r["Calls"], float(r["TotalDurationNs"]) / 1e6 / div,
                                                                         float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
