mkdir -p gpurun_out
(python -m pytest tests/test_ops_gpu.py -q -x -m gpu -k "e4m7 or gelu or epilogue or gemm" > gpurun_out/t7.log 2>&1; echo "rc=$?" >> gpurun_out/t7.log)
(python -m pytest tests/test_model_gpu.py tests/test_dgrad8_gpu.py -q -s -m gpu > gpurun_out/t8.log 2>&1; echo "rc=$?" >> gpurun_out/t8.log)
for v in product bf16gelu product bf16gelu product bf16gelu; do
  if [ "$v" = "bf16gelu" ]; then export CLIBD_GELU_GRAD=bf16; else unset CLIBD_GELU_GRAD; fi
  timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-h2d --no-ref-numerics --no-configs4 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$v', round(d['ms_per_step'],2), 'ms   GEMM frac', round(r['frac'],4), 'gelu_grad', d['config']['numerics']['image_encoder']['gelu_grad'], ' board W', round(r.get('board',{}).get('board_power_w') or 0))" >> gpurun_out/e4m7_ab.log
done
unset CLIBD_GELU_GRAD
cat gpurun_out/e4m7_ab.log; tail -n 4 gpurun_out/t7.log gpurun_out/t8.log
