set -x
mkdir -p gpurun_out/r06
python -m pytest tests/test_gelu_aux8_gpu.py tests/test_gemm_nt4w_gpu.py -x -q -m gpu -s > gpurun_out/r06/tests6.log 2>&1
tail -25 gpurun_out/r06/tests6.log
for i in 1 2; do
WFT_GELU_AUX8=0 python bench.py --no-extras --no-cpu-baseline --no-roofline --steps 8 --warmup 3 --batch 96 > gpurun_out/r06/bench_aux0_$i.log 2>&1
python bench.py --no-extras --no-cpu-baseline --no-roofline --steps 8 --warmup 3 --batch 96 > gpurun_out/r06/bench_aux1_$i.log 2>&1
done
for f in gpurun_out/r06/bench_aux*.log; do echo $f; tail -1 $f | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['step_frac_of_bf16_peak'], d.get('hbm_peak_gib'), d['final_loss'])" || tail -3 $f; done
python -m pytest tests/test_large_v3_gpu.py -x -q -m gpu -k "small_and_medium or batch_of_two" > gpurun_out/r06/tests6b.log 2>&1
tail -15 gpurun_out/r06/tests6b.log
