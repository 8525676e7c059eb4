set -x
mkdir -p gpurun_out/r06
python -m pytest tests/test_attn_prescale_gpu.py tests/test_model_gpu.py -q -m gpu > gpurun_out/r06/tests16.log 2>&1
tail -4 gpurun_out/r06/tests16.log
for i in 1 2; do
WFT_QK_PRESCALE_CROSS=0 python bench.py --no-extras --no-cpu-baseline --no-roofline --steps 8 --warmup 3 > gpurun_out/r06/bench_cross0_$i.log 2>&1
python bench.py --no-extras --no-cpu-baseline --no-roofline --steps 8 --warmup 3 > gpurun_out/r06/bench_cross1_$i.log 2>&1
done
for f in gpurun_out/r06/bench_cross*.log; do echo $f; tail -1 $f | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['step_frac_of_bf16_peak'], d['final_loss'])" || tail -3 $f; done
python -m pytest tests/test_large_v3_gpu.py -q -m gpu > gpurun_out/r06/tests16b.log 2>&1
tail -4 gpurun_out/r06/tests16b.log
