cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/build && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 $R/tools/mfma_peak.hip -o $R/build/mfma_peak && $R/build/mfma_peak > $R/gpurun_out/mfma_peak.log 2>&1
cat $R/gpurun_out/mfma_peak.log
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES --output-format csv -d $R/gpurun_out/pmc2 -- python3 $R/tools/kbench.py --iters 2 > $R/gpurun_out/pmc2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_LDS --output-format csv -d $R/gpurun_out/pmc3 -- python3 $R/tools/kbench.py --iters 2 > $R/gpurun_out/pmc3.log 2>&1
tail -1 $R/gpurun_out/pmc3.log
