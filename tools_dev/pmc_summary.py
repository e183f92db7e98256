#!/usr/bin/env python
"""Per-kernel mean of rocprofv3 --pmc counters from *_counter_collection.csv files.
usage: pmc_summary.py fetch.csv [write.csv ...]"""
import csv
import collections
import sys


def main():
    for path in sys.argv[1:]:
        acc = collections.defaultdict(lambda: [0, 0.0])
        with open(path) as f:
            for row in csv.DictReader(f):
                k = (row.get("Kernel_Name", "?")[:60], row.get("Counter_Name", "?"))
                acc[k][0] += 1
                acc[k][1] += float(row.get("Counter_Value", 0))
        print(f"# {path}")
        for (kern, ctr), (n, tot) in sorted(acc.items()):
            print(f"{kern:60s} {ctr:12s} n={n:4d} mean={tot / n:.1f}")


if __name__ == "__main__":
    main()
