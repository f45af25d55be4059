#!/usr/bin/env python3
"""Condenses a rocprofv3 --kernel-trace --stats output directory into the small CSV kept under
profiles/: every mca:: kernel plus the total of everything else (torch's input generation)."""
import csv
import glob
import sys


def main(src, dst):
    f = sorted(glob.glob(src + "/**/*kernel_stats.csv", recursive=True))[0]
    rows = list(csv.DictReader(open(f)))
    mine = [r for r in rows if "mca" in r["Name"]]
    other = [r for r in rows if "mca" not in r["Name"]]
    with open(dst, "w", newline="") as out:
        w = csv.writer(out)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "StdDev"])
        for r in mine:
            w.writerow([r["Name"], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["MinNs"], r["MaxNs"], r["StdDev"]])
        w.writerow(["(all non-mca kernels: torch synthetic-input generation, copies)", sum(int(r["Calls"]) for r in other),
                    sum(int(r["TotalDurationNs"]) for r in other), "", "", "", ""])
    print(open(dst).read())


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
