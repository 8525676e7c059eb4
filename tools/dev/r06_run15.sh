set -x
mkdir -p gpurun_out/r06
P=whisper-finetune_amd
python -m pytest tests/test_kernels_gpu.py tests/test_attn_dq4w_gpu.py tests/test_attn_dkdv4w_gpu.py tests/test_attn_prescale_gpu.py tests/test_dist_gpu.py tests/test_model_gpu.py -q -m gpu > gpurun_out/r06/tests15.log 2>&1
tail -5 gpurun_out/r06/tests15.log
for i in 1 2; do
WFT_LIB=$PWD/$P/libwft_old.so python bench.py --no-extras --no-cpu-baseline --no-roofline --steps 8 --warmup 3 > gpurun_out/r06/bench_red0_$i.log 2>&1
python bench.py --no-extras --no-cpu-baseline --no-roofline --steps 8 --warmup 3 > gpurun_out/r06/bench_red1_$i.log 2>&1
done
for f in gpurun_out/r06/bench_red*.log; do echo $f; tail -1 $f | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['step_frac_of_bf16_peak'], d['final_loss'])" || tail -3 $f; done
for i in 1 2; do
WFT_LIB=$PWD/$P/libwft_old.so python bench.py --model base --batch 8 --no-extras --no-cpu-baseline --no-roofline --steps 40 --warmup 10 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('base old', d['ms_per_step'], d['ms_per_step_median'])"
python bench.py --model base --batch 8 --no-extras --no-cpu-baseline --no-roofline --steps 40 --warmup 10 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('base new', d['ms_per_step'], d['ms_per_step_median'])"
done
