"""Micro-benchmark of the onesweep passes alone: n (tile|depth)-shaped keys, sorted on the tile
half only (bits [32, 32+tile_bits)), timed with events on the launching stream.
Usage: [GSR_LIB_TAG=variant] python scripts/sort_bench.py [n] [tile_bits] [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from gsrast_amd import _capi
from gsrast_amd.rasterizer import sort_pairs

n = int(sys.argv[1]) if len(sys.argv) > 1 else 267_476_934
tile_bits = int(sys.argv[2]) if len(sys.argv) > 2 else 13
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
mode = os.environ.get("GSR_KEYS", "uniform")
if mode == "const":          # every key in one bin: the scatter degenerates into a copy
    tiles = torch.full((n,), 5, device=dev, dtype=torch.int64)
elif mode == "sorted":       # already sorted by tile: long runs
    tiles = (torch.arange(n, device=dev, dtype=torch.int64) * 8160) // n
else:
    tiles = torch.randint(0, 8160 if tile_bits == 13 else (1 << tile_bits), (n,), device=dev, dtype=torch.int64, generator=g)
depth = torch.randint(0, 1 << 30, (n,), device=dev, dtype=torch.int64, generator=g)
keys = (tiles << 32) | depth
del tiles, depth
vals = torch.arange(n, device=dev, dtype=torch.int32)
out = (torch.empty_like(keys), torch.empty_like(vals))
temp = torch.zeros(int(_capi.lib().gsr_sort_temp_bytes(n)), dtype=torch.uint8, device=dev)
passes = (tile_bits + 7) // 8
for _ in range(2):
    sort_pairs(keys, vals, 32 + tile_bits, 32, out=out, temp=temp)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
ev[0].record()
for _ in range(reps):
    sort_pairs(keys, vals, 32 + tile_bits, 32, sync=False, out=out, temp=temp)
ev[1].record()
torch.cuda.synchronize()
ms = ev[0].elapsed_time(ev[1]) / reps
ok = bool(((out[0][1:] >> 32) >= (out[0][:-1] >> 32)).all())
gbs = n * (8 + 24 * passes) / (ms * 1e-3) / 1e9
print(f"keys={mode} tag={os.environ.get('GSR_LIB_TAG','')} n={n} passes={passes} {ms:.3f} ms  {gbs:.0f} GB/s algorithmic  sorted={ok}")

if os.environ.get("GSR_SHOW_STAMPS"):
    # diagnostic build: 128-byte per-tile slots behind the status words (see radix_sort.hip)
    tiles = (n + 4095) // 4096
    def al(v): return (v + 127) // 128 * 128
    off = al(8 * n) + al(4 * n) + tiles * 256 * 8
    raw = temp[off:off + tiles * 128].view(torch.int64).view(tiles, 16).sum(0).cpu().numpy()
    names = ["ticket", "issue+digitbase", "loadwait+ranking", "totals+publish+scans", "slots", "lookback(d0)",
             "bar(slowest digit)", "write-out", "-"]
    d = tiles * passes * (reps + 2)
    tot = sum(int(raw[i]) for i in range(9))
    for i, nm in enumerate(names):
        print(f"  {nm:22s} {int(raw[i]) / d:10.0f} cycles/tile  {100.0 * int(raw[i]) / max(tot, 1):5.1f} %")
    print(f"  look-back of digit 0: {int(raw[12]) / d:.2f} rounds/tile, {int(raw[13]) / d:.2f} tiles walked/tile, "
          f"{int(raw[14]) / d:.2f} empty rounds/tile; block life {tot / d:.0f} cycles")
