#!/usr/bin/env python3
"""Where a batch-1 call's time goes on the DEVICE: kernel durations and the gaps between the kernels of one call, from a
rocprofv3 --kernel-trace of tools/small_batch_loop.py 1 (timestamps of every dispatch).
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/lat -o p -- python tools/small_batch_loop.py 1
    python tools/latency_gaps.py gpurun_out/lat
A call = 4 dispatches (conv segment kernel, fc.0 GEMV, fc.3 GEMV, tail).  Gap = next kernel's start - this kernel's end."""
import csv, glob, sys
import numpy as np
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].split("<")[0][-40:]))
rows.sort()
rows = rows[-4 * 400:]                      # the last 400 calls: steady state
names = [r[2] for r in rows[:4]]
calls = np.array([[(rows[4 * i + k][0], rows[4 * i + k][1]) for k in range(4)] for i in range(len(rows) // 4)], dtype=np.int64)
dur = (calls[:, :, 1] - calls[:, :, 0]) / 1e3
gap = (calls[:, 1:, 0] - calls[:, :-1, 1]) / 1e3
between = (calls[1:, 0, 0] - calls[:-1, 3, 1]) / 1e3
print("kernels of one call:", names)
print("kernel durations, us (median):", np.round(np.median(dur, 0), 2).tolist(), " sum", round(float(np.median(dur.sum(1))), 2))
print("gaps inside a call, us (median):", np.round(np.median(gap, 0), 2).tolist(), " sum", round(float(np.median(gap.sum(1))), 2))
print("first kernel start -> last kernel end, us (median):", round(float(np.median((calls[:, 3, 1] - calls[:, 0, 0]) / 1e3)), 2))
print("last kernel end -> next call's first kernel start, us (median):", round(float(np.median(between)), 2), "(host: unpack, python, launch)")
print("call period, us (median):", round(float(np.median(np.diff(calls[:, 0, 0]) / 1e3)), 2))
