"""Start and duration of every kernel of ONE frame of a rocprofv3 --kernel-trace run (csv), per queue:
python scripts/frame_timeline.py <trace dir> > profiles/<round>_head_timeline.txt"""
import csv
import glob
import re
import sys

f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "gsr::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
frames, cur = [], []
for r in rows:
    if "preprocess_kernel" in r["Kernel_Name"] and cur:
        frames.append(cur)
        cur = []
    cur.append(r)
frames.append(cur)
fr = frames[len(frames) // 2]
t0 = int(fr[0]["Start_Timestamp"])
queues = {}
print("# One frame of the traced bench command (rocprofv3 --kernel-trace): start and duration of every kernel in microseconds.")
print("# Queue 1 = the caller's stream, queue 2 = the library's second stream: the colours forked behind the preprocess, the tile order")
print("# behind them, the blend beside the emission.")
for r in fr:
    q = queues.setdefault(r["Queue_Id"], len(queues) + 1)
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    m = re.search(r"::(\w+?)(?:<[^(]*>)?\(", r["Kernel_Name"])
    print(f"queue {q}  {(m.group(1) if m else '?'):28s} start {s:8.1f}  duration {e - s:7.1f}  end {e:8.1f}")
print(f"# first kernel's start to last kernel's end: {(max(int(r['End_Timestamp']) for r in fr) - t0) / 1e3:.1f} us")
