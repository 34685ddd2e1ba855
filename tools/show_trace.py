#!/usr/bin/env python3
"""show_trace.py <kernel_trace.csv> [first-frame] [frames]: the kernel schedule of a few frames of the pipelined loop
(start, end, duration in microseconds relative to the first trace kernel shown, queue id), and the trace kernels'
mean period, duration and overlap with their successor."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
first = int(sys.argv[2]) if len(sys.argv) > 2 else 20
n = int(sys.argv[3]) if len(sys.argv) > 3 else 4
def short(nm):
    if "trace_" in nm: return "trace"
    if "rgb8" in nm: return "rgb8"
    if "ccl" in nm.lower() or "llReduce" in nm: return "rccl"
    if "slice_sum" in nm: return "slice_sum"
    return nm[:28]
tr = [r for r in rows if "trace_" in r["Kernel_Name"]]
t0 = int(tr[first]["Start_Timestamp"])
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if t0 <= s <= int(tr[first + n]["End_Timestamp"]):
        print(f"  {short(r['Kernel_Name']):28s} q{r['Queue_Id']:>2} start {(s - t0) / 1e3:8.1f} end {(e - t0) / 1e3:8.1f} dur {(e - s) / 1e3:7.1f} us")
body = tr[8:-2]
st = [int(r["Start_Timestamp"]) for r in body]; en = [int(r["End_Timestamp"]) for r in body]
per = (st[-1] - st[0]) / (len(st) - 1) / 1e3
dur = sum(e - s for s, e in zip(st, en)) / len(st) / 1e3
ov = sum(max(0, en[i] - st[i + 1]) for i in range(len(st) - 1)) / (len(st) - 1) / 1e3
idle = sum(max(0, st[i + 1] - max(en[:i + 1])) for i in range(len(st) - 1)) / (len(st) - 1) / 1e3
print(f"trace kernels: period {per:.1f} us, mean duration {dur:.1f} us, mean overlap with the next {ov:.1f} us, mean gap with no trace kernel running {idle:.1f} us")
