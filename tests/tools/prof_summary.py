#!/usr/bin/env python
"""Summarise a rocprofv3 --kernel-trace CSV: per kernel (and grid size) calls, avg/min us, ms per step."""
import collections, csv, sys
path, steps = sys.argv[1], int(sys.argv[2])
agg = collections.defaultdict(list)
for r in csv.DictReader(open(path)):
    n = r["Kernel_Name"].replace("aimnet::", "").replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
    agg[n[:60]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
tot = sum(sum(v) for v in agg.values())
print(f"total kernel time {tot/1e3:.2f} ms over {steps} steps = {tot/1e3/steps:.3f} ms/step")
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:32]:
    print(f"{k:62s} n/step={len(v)/steps:5.1f} avg={sum(v)/len(v):8.1f}us min={min(v):8.1f} ms/step={sum(v)/1e3/steps:6.3f} {100*sum(v)/tot:5.1f}%")
