"""Reduce two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of `bench.py --steps S --warmup W` to HBM bytes per step.

Follows /opt/skills/guides/MI355X_MICROARCH.md "HBM [CDNA4]": the two counters need separate passes (TCC has 4 slots,
FETCH_SIZE costs 3 and WRITE_SIZE 2); rocprofv3 reports both in KiB; on gfx950 FETCH_SIZE counts 128-B read requests as
64 B, so it is DOUBLED; WRITE_SIZE is exact for 16-B/lane stores and float atomics.  Dispatches are attributed to steps
by counting launches of the once-per-step kernel `sgd_step_kernel` (`sgd_step_seg_kernel` under parameter groups / flat modules); only the last S steps (the timed ones) are kept.

usage: python tools/pmc/pmc_traffic.py <fetch_dir> <write_dir> <steps> <batch> <out.json>
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as ge  # noqa: E402


def per_step(d, counter, steps):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    ends = [i for i, r in enumerate(rows) if "sgd_step_kernel" in r["Kernel_Name"] or "sgd_step_seg_kernel" in r["Kernel_Name"]]
    if len(ends) < steps + 1:
        raise SystemExit(f"{counter}: only {len(ends)} optimiser launches found")
    lo, hi = ends[-steps - 1] + 1, ends[-1] + 1        # the last `steps` complete steps
    fam = defaultdict(float)
    per_step.last_step = [(r["Kernel_Name"].split("(")[0].replace("void ", "")[:80], int(r["Grid_Size"]), float(r["Counter_Value"]) * 1024.0)
                          for r in rows[ends[-2] + 1:hi]]
    for r in rows[lo:hi]:
        name = r["Kernel_Name"]
        key = name.split("(")[0].replace("void ", "")
        key = key.split("<")[0] if not key.startswith("bvc::") else key
        fam[key] += float(r["Counter_Value"]) * 1024.0
    return {k: v / steps for k, v in fam.items()}


def main():
    fetch_dir, write_dir, steps, batch, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
    rd = {k: 2.0 * v for k, v in per_step(fetch_dir, "FETCH_SIZE", steps).items()}     # gfx950 correction
    rd_last = per_step.last_step
    wr = per_step(write_dir, "WRITE_SIZE", steps)
    wr_last = per_step.last_step
    fams = sorted(set(rd) | set(wr), key=lambda k: -(rd.get(k, 0) + wr.get(k, 0)))
    res = {
        "source": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) around bench.py",
        "corrections": "KiB -> bytes; FETCH_SIZE x2 on gfx950 (128-B requests tallied at 64 B)",
        "batch": batch, "steps_averaged": steps,
        "source_hash": ge._source_hash(),      # bench.py reports these bytes only while the library sources are unchanged
        "read_bytes_per_step": sum(rd.values()), "write_bytes_per_step": sum(wr.values()),
        "hbm_bytes_per_step": sum(rd.values()) + sum(wr.values()),
        "by_kernel_GB": {k: {"read": round(rd.get(k, 0) / 1e9, 4), "write": round(wr.get(k, 0) / 1e9, 4)} for k in fams[:40]},
    }
    json.dump(res, open(out, "w"), indent=1)
    if len(rd_last) == len(wr_last):      # per-dispatch table of the last step (same launch sequence in both passes)
        with open(out.replace(".json", "_dispatches.txt"), "w") as f:
            f.write("kernel grid read_MB(x2 corrected) write_MB\n")
            for (n, g, r), (_n2, _g2, w) in zip(rd_last, wr_last):
                f.write(f"{n:80s} {g:9d} {2 * r / 1e6:9.2f} {w / 1e6:9.2f}\n")
    print(json.dumps({k: res[k] for k in ("read_bytes_per_step", "write_bytes_per_step", "hbm_bytes_per_step")}))
    for k in fams[:25]:
        print(f"{k[:90]:90s} rd {rd.get(k, 0)/1e9:8.3f} GB  wr {wr.get(k, 0)/1e9:8.3f} GB")


if __name__ == "__main__":
    main()
