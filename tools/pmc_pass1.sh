cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python3 $R/tools/kbench.py --iters 5 > $R/gpurun_out/kbench.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 --output-format csv -d $R/gpurun_out/pmc1 -- python3 $R/tools/kbench.py --iters 2 > $R/gpurun_out/pmc1.log 2>&1
tail -3 $R/gpurun_out/pmc1.log
cat $R/gpurun_out/kbench.log
