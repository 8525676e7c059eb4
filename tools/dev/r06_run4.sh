set -x
mkdir -p gpurun_out/r06
P=whisper-finetune_amd
for i in 1 2 3; do
for l in libwft.so libwft_fwdm1.so; do for pre in 0 1; do WFT_TIME_PRE=$pre WFT_LIB=$PWD/$P/$l python tools/dev/attn_fwd_time.py; done; done
done > gpurun_out/r06/fwd_pre_ab.log 2>&1
grep -v amdgpu.ids gpurun_out/r06/fwd_pre_ab.log
python -m pytest tests/test_gemm_nt4w_gpu.py tests/test_gemm_tn4w_gpu.py tests/test_grad_homes_gpu.py -x -q -m gpu > gpurun_out/r06/tests4.log 2>&1
tail -5 gpurun_out/r06/tests4.log
for i in 1 2; do
WFT_QK_PRESCALE=0 python bench.py --no-extras --no-cpu-baseline --steps 8 --warmup 3 > gpurun_out/r06/bench_pre0_$i.log 2>&1
python bench.py --no-extras --no-cpu-baseline --steps 8 --warmup 3 > gpurun_out/r06/bench_pre1_$i.log 2>&1
done
for f in gpurun_out/r06/bench_pre*.log; do echo $f; tail -1 $f | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['ms_per_step_median'], d['step_frac_of_bf16_peak'], d['final_loss'])"; done
