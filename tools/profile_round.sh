# usage: bash tools/profile_round.sh <tag>  -- kernel-trace stats of the bench command + HBM traffic counter passes
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${1:-r01}
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_stats -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $R/gpurun_out/${TAG}_stats.log 2>&1
tail -1 $R/gpurun_out/${TAG}_stats.log | cut -c1-160
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${TAG}_fetch -- python3 $R/tools/kbench.py --iters 2 > $R/gpurun_out/${TAG}_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${TAG}_write -- python3 $R/tools/kbench.py --iters 2 > $R/gpurun_out/${TAG}_write.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES --output-format csv -d $R/gpurun_out/${TAG}_sq -- python3 $R/tools/kbench.py --iters 2 > $R/gpurun_out/${TAG}_sq.log 2>&1
python3 $R/tools/pmc_summary.py $R/gpurun_out/${TAG}_sq
python3 - <<PY
import csv, glob, collections
for kind in ("fetch", "write"):
    f = glob.glob("$R/gpurun_out/${TAG}_%s/*/*_counter_collection.csv" % kind)[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"][:60]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        if "svae" in k:
            print("%-6s %-62s avg %.1f (counter units, KB) over %d dispatches" % (kind, k, sum(v) / len(v), len(v)))
PY
