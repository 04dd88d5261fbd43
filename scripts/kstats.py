"""Prints the kernel_stats.csv of a rocprofv3 --kernel-trace --stats --output-format csv run (first match under DIR)."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[: int(sys.argv[2]) if len(sys.argv) > 2 else 40]:
    print(f'{r["Name"][:90]:90s} {r["Calls"]:>6s} {float(r["AverageNs"])/1e3:9.2f} us {r["Percentage"]:>6s}%')
