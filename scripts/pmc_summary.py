"""Per-kernel averages of a rocprofv3 --pmc counter_collection.csv (one row per dispatch and counter)."""
import collections
import csv
import re
import sys


def short(name: str) -> str:
    m = re.search(r"(\w+_kernel(?:<[^>]*>)?)", name)
    if m:
        return m.group(1).replace("unsigned long long", "u64").replace("unsigned int", "u32")
    return name.split("(")[0][-50:]


rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
for r in rows:
    k = short(r["Kernel_Name"])
    acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, d in sorted(acc.items(), key=lambda kv: -sum(dur[kv[0]])):
    n = len(next(iter(d.values())))
    print(f"{k}  dispatches={n} avg_us={sum(dur[k]) / len(dur[k]) / 1e3:.1f}  " +
          " ".join(f"{c}={sum(v) / len(v):.4g}" for c, v in sorted(d.items())))
