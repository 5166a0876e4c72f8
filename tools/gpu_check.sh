#!/bin/bash
# Run on the GPU box (via gpurun): per-kernel parity, whole-step parity, then a short bench.
# A step that is killed by its timeout stops the chain (no further GPU work after a hang).
mkdir -p gpurun_out
rm -f gpurun_out/parity_report.txt
run() {  # name, seconds, command...
  local name=$1 secs=$2; shift 2
  echo "=== $name" | tee -a gpurun_out/summary.txt
  timeout -k 10 "$secs" "$@" > "gpurun_out/$name.log" 2>&1
  local rc=$?
  echo "rc=$rc" | tee -a gpurun_out/summary.txt
  tail -n 15 "gpurun_out/$name.log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name - stopping" | tee -a gpurun_out/summary.txt; exit 1; fi
  return 0
}
: > gpurun_out/summary.txt
rocminfo | grep -E "Marketing Name|Compute Unit|Max Clock" | head -8 >> gpurun_out/summary.txt
for step in "$@"; do
  case $step in
    ops)   run ops 420 python -m pytest tests/test_gpu_ops.py -m gpu -q -p no:cacheprovider ;;
    model) run model 420 python -m pytest tests/test_gpu_videomae.py -m gpu -q -p no:cacheprovider ;;
    all)   run alltests 600 python -m pytest tests -m gpu -q -x -p no:cacheprovider ;;
    smoke) run smoke 200 python -c "import __graft_entry__ as g; g.smoke()" ;;
    bench) run bench 400 python bench.py ;;
    *) echo "unknown step $step" ;;
  esac
done
cat gpurun_out/summary.txt
