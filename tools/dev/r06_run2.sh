set -x
mkdir -p gpurun_out/r06
P=whisper-finetune_amd
for i in 1 2; do
for l in libwft.so libwft_mf16.so libwft_mf16nocs.so libwft_nocs.so; do WFT_LIB=$PWD/$P/$l python tools/dev/attn_bwd_time.py; done
done > gpurun_out/r06/mf16_ab.log 2>&1
cat gpurun_out/r06/mf16_ab.log | grep -v amdgpu.ids
python -m pytest tests/test_headline_sizes_gpu.py -x -q -m gpu > gpurun_out/r06/tests_headline.log 2>&1
tail -30 gpurun_out/r06/tests_headline.log
