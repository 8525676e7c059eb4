set -x
mkdir -p gpurun_out/r06
P=whisper-finetune_amd
python -m pytest tests/test_attn_prescale_gpu.py tests/test_attn_dq4w_gpu.py tests/test_attn_dkdv4w_gpu.py tests/test_attn_fwd_pipe_gpu.py tests/test_gemm_nt4w_gpu.py tests/test_gemm_tn4w_gpu.py tests/test_grad_homes_gpu.py tests/test_headline_sizes_gpu.py -x -q -m gpu > gpurun_out/r06/tests3.log 2>&1
tail -25 gpurun_out/r06/tests3.log
for i in 1 2; do
for l in libwft.so libwft_mf16.so; do for pre in 0 1; do WFT_TIME_PRE=$pre WFT_LIB=$PWD/$P/$l python tools/dev/attn_bwd_time.py; done; done
for pre in 0 1; do WFT_TIME_PRE=$pre python tools/dev/attn_fwd_time.py; done
done > gpurun_out/r06/mf16_ab.log 2>&1
grep -v amdgpu.ids gpurun_out/r06/mf16_ab.log
python -m pytest tests/test_model_gpu.py -x -q -m gpu > gpurun_out/r06/tests3b.log 2>&1
tail -15 gpurun_out/r06/tests3b.log
