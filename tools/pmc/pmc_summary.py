"""Average the counters of the rocprofv3 --pmc passes per (kernel, grid) and print wave-cycle shares.
usage: python tools/pmc/pmc_summary.py <dir_a> <dir_b> <out.txt>"""
import collections, csv, glob, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:3]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "gemm" in r["Kernel_Name"] or "attn" in r["Kernel_Name"]:
                agg[(r["Kernel_Name"].split("(")[0].replace("void ", ""), r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
lines = []
for (k, grid), v in sorted(agg.items()):
    c = {n: sum(x) / len(x) for n, x in v.items()}
    wc = c.get("SQ_WAVE_CYCLES", 0) or 1
    lines.append(f"{k} grid={grid}")
    lines.append("   " + "  ".join(f"{n}={val:.3g}" for n, val in sorted(c.items())))
    if "SQ_WAIT_ANY" in c:
        lines.append(f"   shares of wave cycles: parked (s_waitcnt / barrier) {c['SQ_WAIT_ANY']/wc:.2f}, issue stall {c.get('SQ_WAIT_INST_ANY',0)/wc:.2f}, "
                     f"issuing {c.get('SQ_ACTIVE_INST_ANY',0)/wc:.2f}; MFMA busy / (4 x wave quad-cycles) = {c.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/(4*wc):.2f} per wave")
open(sys.argv[3], "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
