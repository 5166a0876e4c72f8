export TMPDIR=/tmp; mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider > gpurun_out/r05_c_alltests.log 2>&1; echo "tests rc=$?"; tail -n 5 gpurun_out/r05_c_alltests.log
