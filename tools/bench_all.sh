# usage: bash tools/bench_all.sh <tag>  -- one bench.py line per BASELINE config (with cpu_baseline), into gpurun_out/<tag>_bench_cfg<N>.json
cd $GRAFT_REPO_ROOT
TAG=${1:-r03}
for c in 2 1 3 5 4; do
  steps=30; [ $c = 4 ] && steps=10
  timeout -k 10 600 python bench.py --config $c --steps $steps --warmup 3 --cpu-seconds 8 > gpurun_out/${TAG}_bench_cfg$c.log 2>&1
  echo "cfg $c rc=$?"
  grep "^{" gpurun_out/${TAG}_bench_cfg$c.log > gpurun_out/${TAG}_bench_cfg$c.json
  python - <<PY
import json
d=json.load(open("gpurun_out/${TAG}_bench_cfg$c.json"))
r=d["roofline"]; cb=d["cpu_baseline"]
print("  %s: %.1f img/s, %.3f ms/step, dominant %s frac %.3f, step-level mfma frac %.3f, cpu %.1f img/s (%d cores), 1 thread %.1f" % (d["config"]["workload"][:40], d["value"], d["ms_per_step"], r["kernel"], r["frac"], d["config"]["decoder_mfma_frac_of_step"], cb["value"], cb["cores"], cb["one_thread"]["value"]))
PY
done
