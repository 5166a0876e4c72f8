"""MFMA-pipe utilisation of every kernel of one training step, from ONE rocprofv3 --pmc pass over bench.py
(SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CYCLES, GRBM_GUI_ACTIVE, SQ_WAVES - SQ counters only, no TCC slots needed).

  MFMA busy of a kernel = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs)
i.e. the share of the kernel's own GPU cycles in which a SIMD's matrix pipe is executing, averaged over the 1024 SIMDs; the clock the
chip held is GRBM cycles / kernel time when a kernel-trace duration is available.  Dispatches are attributed to steps by the
once-per-step `sgd_step_kernel` / `sgd_step_seg_kernel`; the last S steps are averaged.

usage: python tools/pmc/pmc_mfma_step.py <pmc_dir> <steps> <out.txt>
"""
import csv
import glob
import sys
from collections import defaultdict


def main():
    d, steps, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    disp = defaultdict(dict)
    name = {}
    for r in rows:
        i = int(r["Dispatch_Id"])
        disp[i][r["Counter_Name"]] = float(r["Counter_Value"])
        name[i] = r["Kernel_Name"]
    ids = sorted(disp)
    ends = [k for k, i in enumerate(ids) if "sgd_step_kernel" in name[i] or "sgd_step_seg_kernel" in name[i]]
    if len(ends) < steps + 1:
        raise SystemExit(f"only {len(ends)} optimiser launches found")
    sel = ids[ends[-steps - 1] + 1:ends[-1] + 1]
    fam = defaultdict(lambda: defaultdict(float))
    for i in sel:
        key = name[i].split("(")[0].replace("void ", "")
        key = key.split("<")[0] if not key.startswith("bvc::") else key
        for c, v in disp[i].items():
            fam[key][c] += v / steps
        fam[key]["launches"] += 1.0 / steps
    tot_g = sum(v.get("GRBM_GUI_ACTIVE", 0) for v in fam.values())
    tot_m = sum(v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) for v in fam.values())
    lines = [f"{'kernel':78s} {'launches':>8s} {'GPU Mcyc':>9s} {'share':>6s} {'MFMA busy':>9s}"]
    for k, v in sorted(fam.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0)):
        g = v.get("GRBM_GUI_ACTIVE", 0) / 8.0
        if g <= 0:
            continue
        busy = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024.0 / g
        lines.append(f"{k[:78]:78s} {v['launches']:8.1f} {g / 1e6:9.2f} {100 * v.get('GRBM_GUI_ACTIVE', 0) / tot_g:5.1f}% {busy:9.3f}")
    lines.append(f"{'whole step':78s} {'':8s} {tot_g / 8e6:9.2f} {'100.0%':>6s} {tot_m / 1024.0 / (tot_g / 8.0):9.3f}")
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
