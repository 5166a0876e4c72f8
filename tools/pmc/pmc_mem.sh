#!/bin/bash
# Memory-hierarchy counters of the persistent GEMM on a few products (round 4): where do the L2 -> LDS fills of the K loop come from and
# what do they cost?  Three rocprofv3 --pmc passes per case over tools/pmc/g8_only.py, reduced by tools/pmc/pmc_mem_summary.py.
R=$PWD; OUT=$R/gpurun_out; export TMPDIR=/tmp
rm -rf $OUT/pmcmem; cd /tmp
# BVC_PMC_SCRIPT / BVC_PMC_VAR: another launcher script and the environment variable that names its case (e.g. tools/pmc/dw_only.py, BVC_DW_CASE)
SCRIPT=${BVC_PMC_SCRIPT:-tools/pmc/g8_only.py}; VAR=${BVC_PMC_VAR:-BVC_G8_CASE}
for c in ${BVC_CASES:-square8192 dec_qkv256 enc_qkv256 enc_dxfc1_256}; do
  export $VAR=$c
  n=0
  for set in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_TAG_STALL_sum" \
             "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" \
             "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum"; do
    n=$((n+1))
    # (the counters of one hardware block per pass: more than a block's few slots is refused with "exceeds the capabilities of the hardware")
    timeout -k 10 90 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/pmcmem/$c/p$n -- python3 $R/$SCRIPT >> $OUT/pmcmem_$c.log 2>&1 || { echo "pass $n of $c failed"; tail -3 $OUT/pmcmem_$c.log; exit 1; }
  done
done
cd $R
python3 tools/pmc/pmc_mem_summary.py $OUT/pmcmem $OUT/pmc_mem_summary.txt
find $OUT/pmcmem -name "*.csv" -size +2M -delete
