set -x
mkdir -p gpurun_out/r06
P=whisper-finetune_amd
for i in 1 2 3; do
for l in libwft.so libwft_prio1.so libwft_prio2.so; do WFT_TIME_PRE=1 WFT_LIB=$PWD/$P/$l python tools/dev/attn_fwd_time.py; done
done > gpurun_out/r06/fwd_prio_ab.log 2>&1
grep -v amdgpu.ids gpurun_out/r06/fwd_prio_ab.log
python -m pytest tests/test_gelu_aux8_gpu.py -x -q -m gpu -s > gpurun_out/r06/tests7a.log 2>&1
tail -8 gpurun_out/r06/tests7a.log
python -m pytest tests -x -q -m gpu > gpurun_out/r06/tests7_full.log 2>&1
tail -15 gpurun_out/r06/tests7_full.log
