# usage: [SVAE_GEMM=fp16x3] bash tools/profile_round.sh <tag> [configs, default "2 3 5 4"]
#   per BASELINE config:
#   1. rocprofv3 --kernel-trace --stats of the bench command (the program itself directly behind `--`)
#   2. HBM traffic of the decoder kernels: separate --pmc FETCH_SIZE / WRITE_SIZE passes over tools/kbench.py at the config's shape
#   3. SQ counters (MFMA-busy, wait, instruction mix) over the same workload
# Everything lands under gpurun_out/<tag>_*; copy what is to be judged into profiles/.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r03}
CFGS=${2:-"2 3 5 4"}
MODE=${SVAE_GEMM:-fp32}
for c in $CFGS; do
  steps=20; [ $c = 4 ] && steps=5
  case $c in
    1) KB="--B 64 --n 28 --H 500 --L 2 --z 2 --C 1";;
    2) KB="--B 256 --n 28 --H 500 --L 2 --z 2 --C 1";;
    3) KB="--B 512 --n 40 --H 500 --L 2 --z 2 --C 2";;
    4) KB="--B 128 --n 128 --H 1024 --L 3 --z 20 --C 3";;
    5) KB="--B 256 --n 40 --H 500 --L 2 --z 8 --C 1";;
  esac
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_stats_cfg$c -- python3 $R/bench.py --config $c --steps $steps --warmup 3 --no-cpu-baseline --no-secondary --no-profile --sustained 0 --gemm $MODE > $R/gpurun_out/${TAG}_stats_cfg$c.log 2>&1
  grep "^{" $R/gpurun_out/${TAG}_stats_cfg$c.log | cut -c1-220
  python3 $R/tools/trace_summary.py $R/gpurun_out/${TAG}_stats_cfg$c --steps $steps > $R/gpurun_out/${TAG}_trace_summary_cfg$c.txt 2>&1
  head -8 $R/gpurun_out/${TAG}_trace_summary_cfg$c.txt
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${TAG}_fetch_cfg$c -- python3 $R/tools/kbench.py --iters 2 $KB > $R/gpurun_out/${TAG}_fetch_cfg$c.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${TAG}_write_cfg$c -- python3 $R/tools/kbench.py --iters 2 $KB > $R/gpurun_out/${TAG}_write_cfg$c.log 2>&1
  rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES --output-format csv -d $R/gpurun_out/${TAG}_sq_cfg$c -- python3 $R/tools/kbench.py --iters 2 $KB > $R/gpurun_out/${TAG}_sq_cfg$c.log 2>&1
  python3 $R/tools/pmc_summary.py $R/gpurun_out/${TAG}_sq_cfg$c > $R/gpurun_out/${TAG}_sq_counters_cfg$c.txt 2>&1
  cat $R/gpurun_out/${TAG}_sq_counters_cfg$c.txt
  python3 $R/tools/traffic_json.py $R/gpurun_out/${TAG}_fetch_cfg$c $R/gpurun_out/${TAG}_write_cfg$c $R/gpurun_out/${TAG}_traffic_cfg$c.json "BASELINE cfg $c ($KB)" | grep -E "dense|wgrad|out_bwd|layer0_fwd"
  echo "cfg $c done"
done
