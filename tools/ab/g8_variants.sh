#!/bin/bash
# Same-box A/B of gemm8 K-loop variants: tools/ab/gemm8_ab.py once per library build (BVC_LIB_PATH), product first and last.
R=$PWD; OUT=$R/gpurun_out; mkdir -p $OUT
for v in prod "$@" prod; do
  if [ "$v" = "prod" ]; then unset BVC_LIB_PATH; else export BVC_LIB_PATH=$R/baby-vision-curriculum_amd/libbvc_hip_$v.so; fi
  echo "=== library: $v" | tee -a $OUT/g8_variants.txt
  BVC_ROUNDS=${BVC_ROUNDS:-5} timeout -k 10 300 python tools/ab/gemm8_ab.py 2>/dev/null | grep -v "^tools\|^$" | awk '{print}' | tee -a $OUT/g8_variants.txt || exit 1
done
unset BVC_LIB_PATH
