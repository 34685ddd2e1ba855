#!/usr/bin/env python3
"""summarise_bvh_pmc.py <tag>: the PMC passes tools/pmc_bvh.sh left under gpurun_out/bvh_pmc/ -> profiles/<tag>_bvh_pmc_summary.txt
(values of the LAST dispatch of trace_bvh_kernel in each pass: the timed launch of tools/bench_scene.py)."""
import csv, glob, os, sys
tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
C, lines = {}, [f"# {tag}: rocprofv3 PMC passes (tools/pmc_bvh.sh) over trace_bvh_kernel on the 10,000-disk scene, 1920x1080, 32 spp, depth 8",
                f"# (tools/bench_scene.py --n 10000 --spp 32); values of the timed (second) launch"]
for p in sorted(glob.glob(os.path.join(root, "gpurun_out", "bvh_pmc", "p*"))):
    if not os.path.isdir(p):
        continue
    for f in glob.glob(os.path.join(p, "*", "*_counter_collection.csv")):
        last = {}
        for row in csv.DictReader(open(f)):
            if "trace_bvh_kernel" in row["Kernel_Name"]:
                last[row["Counter_Name"]] = float(row["Counter_Value"])
        for k, v in sorted(last.items()):
            C[k] = v
            lines.append(f"{os.path.basename(p):3s} {k:32s} {v:.5g}")
g = C.get
lines += ["", "# derived"]
if g("GRBM_GUI_ACTIVE"):
    cyc = g("GRBM_GUI_ACTIVE") / 8
    lines.append(f"kernel cycles GRBM_GUI_ACTIVE/8 XCDs: {cyc:.4g}")
    if g("SQ_INSTS_VALU"):
        lines.append(f"VALU issue rate: {g('SQ_INSTS_VALU') / 1024 / cyc:.3f} wave-instructions per cycle per SIMD")
    if g("SQ_WAVE_CYCLES"):
        lines.append(f"mean waves per SIMD 4*SQ_WAVE_CYCLES/(1024*cycles): {4 * g('SQ_WAVE_CYCLES') / 1024 / cyc:.2f}")
    if g("TA_BUSY_avr"):
        lines.append(f"texture-address unit busy TA_BUSY_avr / cycles: {g('TA_BUSY_avr') / cyc:.2f}")
if g("SQ_THREAD_CYCLES_VALU") and g("SQ_ACTIVE_INST_VALU"):
    lines.append(f"lane utilisation SQ_THREAD_CYCLES_VALU/(64*SQ_ACTIVE_INST_VALU): {g('SQ_THREAD_CYCLES_VALU') / (64 * g('SQ_ACTIVE_INST_VALU')):.3f}")
if g("SQ_INSTS_VMEM_RD") and g("SQ_INSTS_VALU"):
    lines.append(f"vector memory reads per VALU instruction: {g('SQ_INSTS_VMEM_RD') / g('SQ_INSTS_VALU'):.3f}; VALU instructions {g('SQ_INSTS_VALU'):.4g}")
open(os.path.join(root, "profiles", f"{tag}_bvh_pmc_summary.txt"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines[-8:]))
