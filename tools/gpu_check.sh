#!/bin/bash
# Run on the GPU box (via gpurun): per-kernel parity, whole-step parity, bench, rocprof.
# Steps micro / gemmdbg / ksweep / racescreen / dwsweep / probe need `exp` first (a -DBVC_EXPERIMENTS build); `noexp` switches back.
# A step that is killed by its timeout stops the chain (no further GPU work after a hang).
R=$PWD
OUT=$R/gpurun_out
mkdir -p $OUT
rm -f $OUT/parity_report.txt
export TMPDIR=/tmp
run() {  # name, seconds, command...
  local name=$1 secs=$2; shift 2
  echo "=== $name" | tee -a $OUT/summary.txt
  timeout -k 10 "$secs" "$@" > "$OUT/$name.log" 2>&1
  local rc=$?
  echo "rc=$rc" | tee -a $OUT/summary.txt
  tail -n 12 "$OUT/$name.log"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "TIMEOUT in $name - stopping" | tee -a $OUT/summary.txt; exit 1; fi
  return 0
}
: > $OUT/summary.txt
rocminfo | grep -E "Marketing Name|Compute Unit|Max Clock" | tail -3 >> $OUT/summary.txt
nproc >> $OUT/summary.txt
for step in "$@"; do
  case $step in
    # the same-process A/B tools flip per-launch environment switches that exist only in an experiments build of the library
    exp)   export BVC_EXTRA_HIPCC_FLAGS=-DBVC_EXPERIMENTS; run expbuild 400 python -c "import __graft_entry__ as g; g.build()" ;;
    noexp) unset BVC_EXTRA_HIPCC_FLAGS; run prodbuild 400 python -c "import __graft_entry__ as g; g.build()" ;;
    g8ab)  run g8ab 500 python tools/ab/gemm8_ab.py ;;
    g8store) run g8store 300 python tools/ab/g8_store_cost.py ;;
    dwwalk) run dwwalk 300 python tools/ab/dw_walk_ab.py ;;
    # forced data-parallel path on ONE GPU (world size 1 over RCCL: every fence, stream hop and collective launch of the N-GPU job,
    # no bytes on xGMI): overhead of the wrapper per bucket size, next to the plain run
    ddpsweep) run ddp_plain 200 python bench.py --no-cpu-baseline --steps 20
              for mb in 5 25 50 100 400; do BVC_FORCE_DDP=1 run ddp_mb$mb 200 python bench.py --no-cpu-baseline --steps 20 --bucket-mb $mb; done
              python3 - <<'PY'
import json, glob, re
rows = []
for f in ["gpurun_out/ddp_plain.log"] + sorted(glob.glob("gpurun_out/ddp_mb*.log"), key=lambda x: int(re.findall(r"mb(\d+)", x)[0])):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        rows.append(f"{f}: unreadable ({e})"); continue
    c = d.get("comm")
    nb = len(c["buckets_last_step"]) if c else 0
    rows.append(f"{f.split('/')[-1]:16s} {d['value']:8.1f} clips/s {d['ms_per_step']:7.3f} ms/step  buckets {nb}" +
                (f"  cap {c['bucket_cap_mb']} MB, per-bucket ms {[b['ms'] for b in c['buckets_last_step']][:4]}..." if c else ""))
open("gpurun_out/ddp_bucket_sweep.txt", "w").write("\n".join(rows) + "\n")
print("\n".join(rows))
PY
              ;;
    ops)   run ops 420 python -m pytest tests/test_gpu_ops.py -m gpu -q -p no:cacheprovider ;;
    ddp)   run ddptests 420 python -m pytest tests/test_gpu_ddp.py -m gpu -q -x -p no:cacheprovider ;;
    model) run model 420 python -m pytest tests/test_gpu_videomae.py -m gpu -q -p no:cacheprovider ;;
    all)   run alltests 900 python -m pytest tests -m gpu -q -x -p no:cacheprovider ;;
    smoke) run smoke 200 python -c "import __graft_entry__ as g; g.smoke()" ;;
    bench) run bench 400 python bench.py ;;
    bench32) run bench32 400 python bench.py --batch 32 --no-cpu-baseline ;;
    diag32) run diag32 300 python bench.py --batch 32 --no-cpu-baseline --per-step
            OMP_NUM_THREADS=4 run diag32omp 300 python bench.py --batch 32 --no-cpu-baseline --per-step
            run diag24 300 python bench.py --batch 24 --no-cpu-baseline --per-step
            run diag48 300 python bench.py --batch 48 --no-cpu-baseline --per-step ;;
    sweepb) for b in 16 32 48 64 96 128; do run bench_b$b 300 python bench.py --batch $b --no-cpu-baseline --steps 20; done ;;
    bench64) run bench64 400 python bench.py --batch 64 --no-cpu-baseline --steps 15 ;;
    benchq) run benchq 300 python bench.py --no-cpu-baseline ;;
    benchddp) BVC_FORCE_DDP=1 run benchddp 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 20 --warmup 3 --no-cpu-baseline ;;
    micro) run micro 400 python tools/ab/microbench.py ;;
    racescreen) run racescreen 500 python tools/persist_race_screen.py ;;
    g8race) run g8race 600 python tools/g8_race_screen.py ;;
    probe_step) run probe_step_b${BVC_BATCH:-256} 300 python tools/step_probe.py ;;
    simclr64) run simclr64 900 python -m pytest tests/test_gpu_simclr.py -m gpu -q -x -s -p no:cacheprovider -k "64_pairs" ;;
    attntests) run attntests 400 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -p no:cacheprovider -k attention ;;
    lossdbg) run lossdbg 300 python tools/debug/g8_loss_dbg.py ;;
    dwab)  run dwab 400 python tools/ab/dw_tile_ab.py ;;
    dwbal) run dwbal 400 python tools/ab/dw_balance_ab.py ;;
    dwhead) BVC_HEAD=1 run dwhead 400 python tools/ab/dw_balance_ab.py ;;
    phase) run phase 500 python tools/ab/phase_ab.py ;;
    attnnw) run attnnw 400 python tools/ab/attn_nw_ab.py ;;
    gemmdbg) run gemmdbg 300 python tools/ab/gemm_dbg.py ;;
    ksweep) run ksweep 400 python tools/ab/gemm_ksweep.py ;;
    dwsweep) run dwsweep 400 python tools/ab/dw_sweep.py ;;
    probe) run probe 300 python tools/gemm_probe.py ;;
    jepa) run jepa 400 python tools/bench_jepa.py ;;
    jepal) run jepal 400 python tools/bench_jepa.py --model vit_large ;;
    simclrvit) run simclrvit 400 python tools/bench_simclr.py --vit ;;
    encode) run encode 300 python tools/bench_encode.py ;;
    simclr) run simclr 400 python tools/bench_simclr.py ;;
    prof)  rm -rf $OUT/prof; cd /tmp
           run prof 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-by-batch --no-probe --no-extra --batch ${BVC_BATCH:-256}
           cd $R
           find $OUT/prof -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats.csv
           find $OUT/prof -name "*kernel_trace.csv" -size +20M -delete ;;
    traffic) rm -rf $OUT/pmct; cd /tmp
           run traffic_rd 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmct/rd -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-by-batch --no-probe --no-extra --batch ${BVC_BATCH:-256}
           run traffic_wr 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmct/wr -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-by-batch --no-probe --no-extra --batch ${BVC_BATCH:-256}
           cd $R
           run traffic 60 python tools/pmc/pmc_traffic.py $OUT/pmct/rd $OUT/pmct/wr 3 ${BVC_BATCH:-256} $OUT/traffic_b${BVC_BATCH:-256}.json
           find $OUT/pmct -name "*.csv" -size +5M -delete ;;
    mfma)  rm -rf $OUT/pmcm; cd /tmp
           run mfma_pmc 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES --output-format csv -d $OUT/pmcm -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-by-batch --no-probe --no-extra --batch ${BVC_BATCH:-256}
           cd $R
           run mfma_table 60 python tools/pmc/pmc_mfma_step.py $OUT/pmcm 3 $OUT/mfma_busy_b${BVC_BATCH:-256}.txt
           find $OUT/pmcm -name "*.csv" -size +5M -delete ;;
    profdefault) rm -rf $OUT/profd; cd /tmp
           run profdefault 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/profd -- python3 $R/bench.py
           cd $R; find $OUT/profd -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats_default.csv
           find $OUT/profd -name "*kernel_trace.csv" -delete ;;
    profnobb) rm -rf $OUT/profn; cd /tmp      # the default command without the 64- / 16-clip legs: every launch of a kernel is at the headline batch
           run profnobb 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/profn -- python3 $R/bench.py --no-by-batch --no-extra
           cd $R; find $OUT/profn -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats_no_by_batch.csv
           find $OUT/profn -name "*kernel_trace.csv" -delete ;;
    prof_jepa) rm -rf $OUT/prof_jepa; cd /tmp
           run prof_jepa 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_jepa -- python3 $R/tools/bench_jepa.py --model vit_large --steps 5 --warmup 2
           cd $R; find $OUT/prof_jepa -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats_jepa_vitl.csv
           find $OUT/prof_jepa -name "*kernel_trace.csv" -delete ;;
    prof_simclr) rm -rf $OUT/prof_simclr; cd /tmp
           run prof_simclr 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_simclr -- python3 $R/tools/bench_simclr.py --vit
           cd $R; find $OUT/prof_simclr -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/kernel_stats_simclr_vitb.csv
           find $OUT/prof_simclr -name "*kernel_trace.csv" -delete ;;
    table) run table 60 python tools/pmc/roofline_table.py $OUT/kernel_stats.csv $OUT/traffic_b${BVC_BATCH:-256}.json 7 $OUT/roofline_table_b${BVC_BATCH:-256}.txt ;;
    pmc_gemm) rm -rf $OUT/pmcg; cd /tmp
           run pmc_gemm1 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA --output-format csv -d $OUT/pmcg/a -- python3 $R/tools/pmc/gemm_only.py
           run pmc_gemm2 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/pmcg/b -- python3 $R/tools/pmc/gemm_only.py
           cd $R
           run pmc_gemm_summary 60 python tools/pmc/pmc_summary.py $OUT/pmcg/a $OUT/pmcg/b $OUT/pmc_gemm_summary.txt ;;
    pmc_g8) rm -rf $OUT/pmcg8; cd /tmp
           for c in square8192 square8192_bn128 dec_qkv enc_fc1 square8192_128x128; do
             export BVC_G8_CASE=$c
             run pmc_g8_${c}_a 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA --output-format csv -d $OUT/pmcg8/$c/a -- python3 $R/tools/pmc/g8_only.py
             run pmc_g8_${c}_b 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/pmcg8/$c/b -- python3 $R/tools/pmc/g8_only.py
             run pmc_g8_${c}_c 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmcg8/$c/c -- python3 $R/tools/pmc/g8_only.py
             python3 $R/tools/pmc/pmc_summary.py $OUT/pmcg8/$c/a $OUT/pmcg8/$c/b $OUT/pmc_g8_$c.txt > /dev/null
             python3 $R/tools/pmc/pmc_summary.py $OUT/pmcg8/$c/c $OUT/pmcg8/$c/c $OUT/pmc_g8_${c}_c.txt > /dev/null
             find $OUT/pmcg8/$c -name "*.csv" -size +2M -delete
           done
           unset BVC_G8_CASE; cd $R; cat $OUT/pmc_g8_*.txt ;;
    pmc_dw) rm -rf $OUT/pmcdw; cd /tmp
           for c in dec10 dec12 enc10; do
             export BVC_DW_CASE=$c
             run pmc_dw_${c}_a 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA --output-format csv -d $OUT/pmcdw/$c/a -- python3 $R/tools/pmc/dw_only.py
             run pmc_dw_${c}_c 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmcdw/$c/c -- python3 $R/tools/pmc/dw_only.py
             python3 $R/tools/pmc/pmc_summary.py $OUT/pmcdw/$c/a $OUT/pmcdw/$c/c $OUT/pmc_dw_$c.txt > /dev/null
             find $OUT/pmcdw/$c -name "*.csv" -size +2M -delete
           done
           unset BVC_DW_CASE; cd $R; cat $OUT/pmc_dw_*.txt ;;
    pmc_attn) rm -rf $OUT/pmcattn; cd /tmp
           run pmc_attn_a 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA --output-format csv -d $OUT/pmcattn/a -- python3 $R/tools/pmc/attn_only.py
           run pmc_attn_b 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INST_CYCLES_VMEM --output-format csv -d $OUT/pmcattn/b -- python3 $R/tools/pmc/attn_only.py
           run pmc_attn_c 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmcattn/c -- python3 $R/tools/pmc/attn_only.py
           python3 $R/tools/pmc/pmc_summary.py $OUT/pmcattn/a $OUT/pmcattn/b $OUT/pmc_attn.txt > /dev/null
           python3 $R/tools/pmc/pmc_summary.py $OUT/pmcattn/c $OUT/pmcattn/c $OUT/pmc_attn_c.txt > /dev/null
           find $OUT/pmcattn -name "*.csv" -size +2M -delete
           cd $R; cat $OUT/pmc_attn.txt $OUT/pmc_attn_c.txt ;;
    attnab) for v in $BVC_VARIANTS; do
             if [ "$v" = "prod" ]; then unset BVC_LIB_PATH; else export BVC_LIB_PATH=$R/baby-vision-curriculum_amd/libbvc_hip_$v.so; fi
             run attnab_$v 200 python tools/ab/attn_ab.py
           done; unset BVC_LIB_PATH ;;
    *) echo "unknown step $step" ;;
  esac
done
cat $OUT/summary.txt
