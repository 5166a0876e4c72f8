#!/usr/bin/env python3
"""One entry point for the measurement tools of this repository.

    python tools/bvc_tools.py list                       # what exists, one line each
    python tools/bvc_tools.py ab <name> [args ...]       # a same-process A/B or experiment (tools/ab/<name>.py or <name>_ab.py)
    python tools/bvc_tools.py pmc <name> [args ...]      # a counter reduction / single-kernel driver (tools/pmc/<name>.py or pmc_<name>.py)
    python tools/bvc_tools.py bench <jepa|simclr|encode> [args ...]
    python tools/bvc_tools.py screen [gemm8|persist]     # race screens
    python tools/bvc_tools.py probe [step|gemm]          # per-product tables of one training step

Every sub-command runs the script of that name unchanged (its own docstring says what it measures and which profile files it
produced); environment variables the scripts read (BVC_BATCH, BVC_ROUNDS, BVC_GEMM_DEBUG ...) pass through.  Sequences of steps on a
GPU box - build, tests, bench, rocprofv3 passes - are `tools/gpu_check.sh <step> ...`."""
import glob
import os
import runpy
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
GROUPS = {"ab": ("ab", ("{}.py", "{}_ab.py")), "pmc": ("pmc", ("{}.py", "pmc_{}.py"))}
TOP = {"bench": {"jepa": "bench_jepa.py", "simclr": "bench_simclr.py", "encode": "bench_encode.py"},
       "screen": {"gemm8": "g8_race_screen.py", "persist": "persist_race_screen.py"},
       "probe": {"step": "step_probe.py", "gemm": "gemm_probe.py"}}


def _first_doc_line(path):
    try:
        src = open(path).read()
    except OSError:
        return ""
    for q in ('"""', "'''"):
        i = src.find(q)
        if 0 <= i < 200:
            return " ".join(src[i + 3:src.find(q, i + 3)].split())[:150]
    return ""


def _list():
    for group, (sub, _pats) in GROUPS.items():
        print(f"{group}:")
        for f in sorted(glob.glob(os.path.join(HERE, sub, "*.py"))):
            n = os.path.basename(f)[:-3]
            if n != "__init__":
                print(f"  {n:22s} {_first_doc_line(f)}")
    for group, names in TOP.items():
        print(f"{group}:")
        for n, f in names.items():
            print(f"  {n:22s} {_first_doc_line(os.path.join(HERE, f))}")


def _resolve(group, name):
    if group in GROUPS:
        sub, pats = GROUPS[group]
        for p in pats:
            f = os.path.join(HERE, sub, p.format(name))
            if os.path.exists(f):
                return f
    elif group in TOP and name in TOP[group]:
        return os.path.join(HERE, TOP[group][name])
    return None


def main():
    if len(sys.argv) < 2 or sys.argv[1] in ("-h", "--help", "help"):
        print(__doc__)
        return 0
    if sys.argv[1] == "list":
        _list()
        return 0
    group = sys.argv[1]
    default = {"screen": "gemm8", "probe": "step"}.get(group)
    name = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("-") else default
    path = _resolve(group, name) if name else None
    if path is None:
        print(f"unknown tool '{group} {name}'; try: python tools/bvc_tools.py list", file=sys.stderr)
        return 2
    rest = sys.argv[3:] if len(sys.argv) > 2 and sys.argv[2] == name else sys.argv[2:]
    sys.argv = [path] + rest
    runpy.run_path(path, run_name="__main__")
    return 0


if __name__ == "__main__":
    sys.exit(main())
