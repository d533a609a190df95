# usage: bash tools/configs_ab.sh -- decoder fwd+bwd kernel time at the BASELINE configs, fp32 MFMA vs fp16x3 (scratch output)
for cfg in "cfg2 --B 256 --n 28 --H 500 --L 2 --z 2 --C 1" "cfg3 --B 512 --n 40 --H 500 --L 2 --z 2 --C 2" "cfg4 --B 128 --n 128 --H 1024 --L 3 --z 20 --C 3" "cfg5 --B 256 --n 40 --H 500 --L 2 --z 8 --C 1"; do
  set -- $cfg; name=$1; shift
  for mode in fp32 fp16x3; do
    if [ $mode = fp16x3 ]; then export SVAE_GEMM=fp16x3; else unset SVAE_GEMM; fi
    timeout -k 10 200 python tools/kbench.py "$@" --iters 5 > gpurun_out/kb_${name}_$mode.log 2>&1
    echo "$name $mode: $(grep -E 'total|TOTAL|sum' gpurun_out/kb_${name}_$mode.log | tail -1)"
  done
done
