"""Reduce the passes of tools/pmc/pmc_mem.sh: per case (directory) the counters of the GEMM kernel averaged per dispatch, plus derived
figures - L2 hit rate, average L1 -> L2 read latency, texture-addresser busy share."""
import collections, csv, glob, os, sys
root, out = sys.argv[1], sys.argv[2]
lines = []
for case in sorted(os.listdir(root)):
    agg = collections.defaultdict(list)
    kern = None
    for f in glob.glob(os.path.join(root, case, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if "gemm" in r["Kernel_Name"]:
                kern = r["Kernel_Name"].split("(")[0].replace("void ", "")
                agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    c = {n: sum(x) / len(x) for n, x in agg.items()}
    lines.append(f"{case}: {kern}")
    lines.append("   " + "  ".join(f"{n}={v:.4g}" for n, v in sorted(c.items())))
    d = []
    if c.get("TCC_REQ_sum"):
        d.append(f"L2 hit rate {c.get('TCC_HIT_sum', 0) / (c.get('TCC_HIT_sum', 0) + c.get('TCC_MISS_sum', 1e-9)):.3f}")
    if c.get("TCP_TCC_READ_REQ_sum"):
        d.append(f"mean L1->L2 read latency {c.get('TCP_TCC_READ_REQ_LATENCY_sum', 0) / c['TCP_TCC_READ_REQ_sum']:.0f} cycles")
    if "TA_BUSY_avr" in c:
        d.append(f"TA busy {c['TA_BUSY_avr']:.1f} %")
    if c.get("GRBM_GUI_ACTIVE"):
        d.append(f"GPU cycles {c['GRBM_GUI_ACTIVE']:.3g}")
    lines.append("   derived: " + ", ".join(d))
open(out, "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
