set -x
mkdir -p gpurun_out/r06
python -m pytest tests/test_launch_mode_gpu.py tests/test_attn_prescale_gpu.py tests/test_attn_fwd_pipe_gpu.py -x -q -m gpu > gpurun_out/r06/tests5.log 2>&1
tail -15 gpurun_out/r06/tests5.log
for b in 87 96 87 96 93; do
python bench.py --no-extras --no-cpu-baseline --no-roofline --steps 8 --warmup 3 --batch $b > gpurun_out/r06/bench_b${b}_$RANDOM.log 2>&1
done
for f in gpurun_out/r06/bench_b*.log; do echo $f; tail -1 $f | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['step_frac_of_bf16_peak'], d.get('hbm_peak_gib'))" || tail -3 $f; done
python -m pytest tests/test_model_gpu.py -x -q -m gpu -k "small_and_medium" > gpurun_out/r06/tests5b.log 2>&1
tail -15 gpurun_out/r06/tests5b.log
