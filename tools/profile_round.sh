# usage: [SVAE_GEMM=fp16x3] bash tools/profile_round.sh <tag>
#   1. rocprofv3 --kernel-trace --stats of the bench command at BASELINE configs 2 (headline), 3, 4, 5
#   2. HBM traffic of the decoder kernels at config 2: separate --pmc FETCH_SIZE / WRITE_SIZE passes over tools/kbench.py
#   3. SQ counters (MFMA-busy, wait, instruction mix) over the same workload
# Everything lands under gpurun_out/<tag>_*; copy what is to be judged into profiles/.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r02}
MODE=${SVAE_GEMM:-fp32}
for c in 2 3 5 4; do
  steps=20; [ $c = 4 ] && steps=5
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_stats_cfg$c -- python3 $R/bench.py --config $c --steps $steps --warmup 3 --no-cpu-baseline --no-secondary --no-profile --sustained 0 --gemm $MODE > $R/gpurun_out/${TAG}_stats_cfg$c.log 2>&1
  grep "^{" $R/gpurun_out/${TAG}_stats_cfg$c.log | cut -c1-220
  python3 $R/tools/trace_summary.py $R/gpurun_out/${TAG}_stats_cfg$c --steps $steps > $R/gpurun_out/${TAG}_trace_summary_cfg$c.txt 2>&1
  head -8 $R/gpurun_out/${TAG}_trace_summary_cfg$c.txt
done
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${TAG}_fetch -- python3 $R/tools/kbench.py --iters 2 > $R/gpurun_out/${TAG}_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${TAG}_write -- python3 $R/tools/kbench.py --iters 2 > $R/gpurun_out/${TAG}_write.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES --output-format csv -d $R/gpurun_out/${TAG}_sq -- python3 $R/tools/kbench.py --iters 2 > $R/gpurun_out/${TAG}_sq.log 2>&1
python3 $R/tools/pmc_summary.py $R/gpurun_out/${TAG}_sq > $R/gpurun_out/${TAG}_sq_counters.txt 2>&1
cat $R/gpurun_out/${TAG}_sq_counters.txt
python3 $R/tools/traffic_json.py $R/gpurun_out/${TAG}_fetch $R/gpurun_out/${TAG}_write $R/gpurun_out/${TAG}_traffic.json
