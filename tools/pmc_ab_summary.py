#!/usr/bin/env python3
"""pmc_ab_summary.py <dir>: per-launch means of the counters tools/pmc_ab.sh collected, one line per
(library, spp), with the figures derived from them (VALU instructions per 64-segment trip, cycles per trip per
SIMD, waves per SIMD, lane utilisation)."""
import csv, glob, os, re, sys
d = sys.argv[1]
SEGS = {1024: 6549558725, 128: 818776325}       # pt_segments of the default scene, 1080p, depth 8, seed 0 (tools/kbench)
rows = {}
for p in sorted(glob.glob(os.path.join(d, "*_*_*"))):
    m = re.match(r"(.+)_(\d+)_(sq|grbm)$", os.path.basename(p))
    if not m or not os.path.isdir(p):
        continue
    key = (m.group(1), int(m.group(2)))
    for f in glob.glob(os.path.join(p, "*", "*_counter_collection.csv")):
        acc = {}
        for row in csv.DictReader(open(f)):
            if "trace_static_kernel" in row["Kernel_Name"]:
                acc.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
        for k, v in acc.items():
            v = v[1:] if len(v) > 1 else v       # the first launch of a process is cold
            rows.setdefault(key, {})[k] = sum(v) / len(v)
for (name, spp), c in sorted(rows.items(), key=lambda kv: (kv[0][1], kv[0][0])):
    trips = SEGS.get(spp, 0) / 64
    cyc = c.get("GRBM_GUI_ACTIVE", 0) / 8
    out = [f"{name:22s} spp={spp:5d}"]
    if "SQ_INSTS_VALU" in c and trips:
        out.append(f"VALU/trip {c['SQ_INSTS_VALU'] / trips:7.2f}  SALU/trip {c.get('SQ_INSTS_SALU', 0) / trips:6.2f}  LDS/trip {c.get('SQ_INSTS_LDS', 0) / trips:5.2f}")
    if cyc and trips:
        out.append(f"cycles {cyc:.4g}  cycles/trip/SIMD {cyc / (trips / 1024):7.2f}")
    if cyc and "SQ_WAVE_CYCLES" in c:
        out.append(f"waves/SIMD {4 * c['SQ_WAVE_CYCLES'] / 1024 / cyc:5.2f}")
    if "SQ_THREAD_CYCLES_VALU" in c and "SQ_ACTIVE_INST_VALU" in c:
        out.append(f"lane util {c['SQ_THREAD_CYCLES_VALU'] / (64 * c['SQ_ACTIVE_INST_VALU']):.3f}")
    print("  ".join(out))
