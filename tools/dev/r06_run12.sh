set -x
mkdir -p gpurun_out/r06
P=whisper-finetune_amd
python -m pytest tests/test_gelu_aux8_gpu.py tests/test_gemm_nt4w_gpu.py tests/test_kernels_gpu.py -q -m gpu > gpurun_out/r06/tests12.log 2>&1
tail -6 gpurun_out/r06/tests12.log
for i in 1 2; do
WFT_LIB=$PWD/$P/libwft_nopoly.so python bench.py --no-extras --no-cpu-baseline --no-roofline --steps 8 --warmup 3 > gpurun_out/r06/bench_poly0_$i.log 2>&1
python bench.py --no-extras --no-cpu-baseline --no-roofline --steps 8 --warmup 3 > gpurun_out/r06/bench_poly1_$i.log 2>&1
done
for f in gpurun_out/r06/bench_poly*.log; do echo $f; tail -1 $f | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['value'], d['ms_per_step'], d['step_frac_of_bf16_peak'], d['final_loss'])" || tail -3 $f; done
python -m pytest tests/test_model_gpu.py tests/test_large_v3_gpu.py -q -m gpu -k "not lora_muon" > gpurun_out/r06/tests12b.log 2>&1
tail -6 gpurun_out/r06/tests12b.log
