# usage: bash tools/pmc_pass.sh <tag>   -- two counter passes over tools/kbench.py, summaries to gpurun_out/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-x}
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES --output-format csv -d $R/gpurun_out/pmc_${TAG}_a -- python3 $R/tools/kbench.py --iters 2 > $R/gpurun_out/pmc_${TAG}_a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU --output-format csv -d $R/gpurun_out/pmc_${TAG}_b -- python3 $R/tools/kbench.py --iters 2 > $R/gpurun_out/pmc_${TAG}_b.log 2>&1
python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_${TAG}_a $R/gpurun_out/pmc_${TAG}_b
