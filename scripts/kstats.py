"""print the top rows of a rocprofv3 kernel_stats.csv with the kernel names cut short: python scripts/kstats.py FILE [rows]"""
import csv, sys
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 24
for i, r in enumerate(csv.DictReader(open(sys.argv[1]))):
    if i < rows:
        print(r["Name"][:72].ljust(72), r["Calls"].rjust(6), ("%.1f" % (float(r["AverageNs"]) / 1e3)).rjust(9), "us", r["Percentage"].rjust(7), "%")
