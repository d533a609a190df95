# usage: bash tools/split_ablate.sh -- time the fp16x3 forward kernel with the ablation builds under build/
export SVAE_GEMM=fp16x3
for v in main ab1 ab2 ab3; do
  cp build/libsvae_$v.so spatial_vae_amd/libsvae_hip.so
  timeout -k 10 200 python bench.py --gemm fp16x3 --no-secondary --no-cpu-baseline --steps 20 > gpurun_out/bench_$v.json 2> gpurun_out/bench_$v.err
  python - <<PY
import json
d = json.load(open("gpurun_out/bench_$v.json"))
k = d["roofline"]["kernels_ms_per_step"]
print("$v", d["ms_per_step"], "dense_fwd", k["dense_fwd"], "prepare", k["prepare"])
PY
done
