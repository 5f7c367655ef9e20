#!/usr/bin/env python3
"""Timeline digest of a rocprofv3 --kernel-trace CSV: how busy the GPU was and how much kernels of different queues overlapped.
Usage: tools/timeline.py trace_kernel_trace.csv [from_fraction [to_fraction]]   (window of the run, as fractions of its dispatches)"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("rt::", ""), r.get("Queue_Id", "?")) for r in rows))
to = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
ev = ev[int(len(ev) * skip):int(len(ev) * to)]
t0, t1 = ev[0][0], max(e[1] for e in ev)
pts = []
for s, e, n, q in ev:
    pts.append((s, 1)); pts.append((e, -1))
pts.sort()
depth_time = collections.Counter()
d, last = 0, t0
for t, k in pts:
    depth_time[d] += t - last
    last = t; d += k
span = t1 - t0
print("span %.3f ms, %d dispatches, queues %s" % (span / 1e6, len(ev), sorted(set(e[3] for e in ev))))
for k in sorted(depth_time):
    print("  %d kernels running: %6.3f ms (%.1f %%)" % (k, depth_time[k] / 1e6, 100.0 * depth_time[k] / span))
by = collections.defaultdict(lambda: [0, 0])
for s, e, n, q in ev:
    by[n][0] += 1; by[n][1] += e - s
for n, (c, t) in sorted(by.items(), key=lambda x: -x[1][1]):
    print("  %-28s %5d x %8.1f us = %8.3f ms (%.1f %% of span)" % (n[:28], c, t / c / 1e3, t / 1e6, 100.0 * t / span))
# gaps inside a queue: end of a dispatch to start of the next one on the same queue
gaps = collections.defaultdict(list)
lastq = {}
for s, e, n, q in ev:
    if q in lastq: gaps[q].append(s - lastq[q])
    lastq[q] = e
for q, g in gaps.items():
    g.sort()
    print("  queue %s: %d gaps, median %.1f us, mean %.1f us, sum %.3f ms" % (q, len(g), g[len(g) // 2] / 1e3, sum(g) / len(g) / 1e3, sum(g) / 1e6))
