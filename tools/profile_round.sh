# usage: [SVAE_GEMM=fp16x3] bash tools/profile_round.sh <tag>  -- kernel-trace stats of the bench command + HBM traffic counter passes
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r01}
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_stats -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary --gemm ${SVAE_GEMM:-fp32} > $R/gpurun_out/${TAG}_stats.log 2>&1
tail -1 $R/gpurun_out/${TAG}_stats.log | cut -c1-160
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${TAG}_fetch -- python3 $R/tools/kbench.py --iters 2 > $R/gpurun_out/${TAG}_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${TAG}_write -- python3 $R/tools/kbench.py --iters 2 > $R/gpurun_out/${TAG}_write.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES --output-format csv -d $R/gpurun_out/${TAG}_sq -- python3 $R/tools/kbench.py --iters 2 > $R/gpurun_out/${TAG}_sq.log 2>&1
python3 $R/tools/pmc_summary.py $R/gpurun_out/${TAG}_sq
python3 $R/tools/traffic_json.py $R/gpurun_out/${TAG}_fetch $R/gpurun_out/${TAG}_write $R/gpurun_out/${TAG}_traffic.json
