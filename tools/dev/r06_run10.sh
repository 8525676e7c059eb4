set -x
mkdir -p gpurun_out/r06
python -m pytest tests/test_model_gpu.py -q -m gpu -k "deferred or train_step or loss_curve" > gpurun_out/r06/tests10.log 2>&1
tail -4 gpurun_out/r06/tests10.log
for i in 1 2; do
WFT_DEFER_LOSS=0 python bench.py --no-extras --no-cpu-baseline --no-roofline --steps 10 --warmup 3 > gpurun_out/r06/bench_defer0_$i.log 2>&1
python bench.py --no-extras --no-cpu-baseline --no-roofline --steps 10 --warmup 3 > gpurun_out/r06/bench_defer1_$i.log 2>&1
done
for f in gpurun_out/r06/bench_defer*.log; do echo $f; tail -1 $f | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['ms_per_step_median'], d['step_frac_of_bf16_peak'], d['final_loss'])" || tail -3 $f; done
