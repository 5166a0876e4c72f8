#!/bin/bash
# Same-box A/B of library variants (tools/build_variant.sh): runs "$BVC_CMD" once per library, the product library first and last.
# usage: BVC_CMD="python tools/ab/g8_tiles_ab.py" tools/ab/lib_cmp.sh <variant> [<variant> ...]
R=$PWD
for v in prod "$@" prod; do
  if [ "$v" = "prod" ]; then unset BVC_LIB_PATH; else export BVC_LIB_PATH=$R/baby-vision-curriculum_amd/libbvc_hip_$v.so; fi
  echo "=== library: $v"
  timeout -k 10 ${BVC_CMD_TIMEOUT:-300} $BVC_CMD 2>&1 | grep -v "amdgpu.ids" || exit 1
done
