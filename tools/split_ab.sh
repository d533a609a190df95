# usage: bash tools/split_ab.sh  -- parity subset + bench in fp16x3 mode (scratch output under gpurun_out/)
export SVAE_GEMM=fp16x3
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_geometry.py -x -q -m gpu > gpurun_out/gpu_tests_split.log 2>&1
tail -3 gpurun_out/gpu_tests_split.log
timeout -k 10 200 python bench.py --gemm fp16x3 --no-secondary --no-cpu-baseline > gpurun_out/bench_split.json 2> gpurun_out/bench_split.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/bench_split.json"))
print(d["value"], d["ms_per_step"])
print(d["roofline"]["kernels_ms_per_step"])
PY
