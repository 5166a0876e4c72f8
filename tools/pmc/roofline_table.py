"""Join the rocprofv3 kernel-stats CSV (time) with tools/pmc/pmc_traffic.py's JSON (fabric bytes) into one per-kernel table:
us/step, GB/step, achieved TB/s.  A kernel near ~5 TB/s is bandwidth-bound whatever its MFMA share; one far below both
rooflines is latency/occupancy-bound.

usage: python tools/pmc/roofline_table.py <kernel_stats.csv> <traffic.json> <profiled_steps> <out.txt>
"""
import csv
import json
import sys
from collections import defaultdict


def fam(name):
    key = name.split("(")[0].replace("void ", "")
    return key.split("<")[0] if not key.startswith("bvc::") else key


def main():
    stats, traffic, steps, out = sys.argv[1], json.load(open(sys.argv[2])), int(sys.argv[3]), sys.argv[4]
    t = defaultdict(float)
    calls = defaultdict(float)
    for r in csv.DictReader(open(stats)):
        t[fam(r["Name"])] += float(r["TotalDurationNs"]) / steps / 1e3
        calls[fam(r["Name"])] += float(r["Calls"]) / steps
    by = traffic["by_kernel_GB"]
    rows = []
    for k, us in t.items():
        b = by.get(k, {"read": 0.0, "write": 0.0})
        gb = b["read"] + b["write"]
        rows.append((us, k, calls[k], b["read"], b["write"], gb / (us * 1e-6) / 1e3 if us > 0 else 0.0))
    rows.sort(reverse=True)
    tot_us = sum(r[0] for r in rows)
    lines = [f"{'kernel':78s} {'calls':>6s} {'us/step':>9s} {'%':>5s} {'rd GB':>7s} {'wr GB':>7s} {'TB/s':>6s}"]
    for us, k, c, rd, wr, tbs in rows:
        if us < 1.0:
            continue
        lines.append(f"{k[:78]:78s} {c:6.1f} {us:9.1f} {100 * us / tot_us:5.1f} {rd:7.3f} {wr:7.3f} {tbs:6.2f}")
    lines.append(f"{'total':78s} {'':6s} {tot_us:9.1f} {100.0:5.1f} {traffic['read_bytes_per_step'] / 1e9:7.3f} "
                 f"{traffic['write_bytes_per_step'] / 1e9:7.3f} {traffic['hbm_bytes_per_step'] / tot_us / 1e6:6.2f}")
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
