#!/bin/bash
# Same-box A/B of library builds: `tools/ab/lib_ab.sh <variant> [bench args]` runs bench.py twice with the product library and twice with
# baby-vision-curriculum_amd/libbvc_hip_<variant>.so (BVC_LIB_PATH), interleaved.
R=$PWD; v=$1; shift
for i in 1 2; do
  for w in prod $v; do
    if [ "$w" = "prod" ]; then unset BVC_LIB_PATH; else export BVC_LIB_PATH=$R/baby-vision-curriculum_amd/libbvc_hip_$w.so; fi
    timeout -k 10 300 python bench.py --no-cpu-baseline --steps 15 "$@" 2>/dev/null | python3 -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print('$w', d['value'], d['ms_per_step'])" || exit 1
  done
done
unset BVC_LIB_PATH
