"""Compact per-kernel table of a rocprofv3 --kernel-trace --stats run (the *_kernel_stats.csv)."""
import csv
import re
import sys


def short(name: str) -> str:
    m = re.search(r"(\w+_kernel(?:<[^>]*>)?|__amd\w+)", name)
    return (m.group(1) if m else name.split("(")[0][-48:]).replace("unsigned long long", "u64").replace("unsigned int", "u32")


if __name__ == "__main__":
    rows = list(csv.DictReader(open(sys.argv[1])))
    limit = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    for r in rows[:limit]:
        print(f"{short(r['Name']):44s} calls={r['Calls']:>5s} avg_us={float(r['AverageNs']) / 1e3:9.1f} "
              f"min_us={float(r['MinNs']) / 1e3:9.1f} max_us={float(r['MaxNs']) / 1e3:9.1f}  {float(r['Percentage']):5.1f} %")
