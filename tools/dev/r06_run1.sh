set -x
mkdir -p gpurun_out/r06
P=whisper-finetune_amd
for i in 1 2; do
for l in libwft.so libwft_nocs.so; do WFT_LIB=$PWD/$P/$l python tools/dev/attn_bwd_time.py; WFT_LIB=$PWD/$P/$l python tools/dev/attn_fwd_time.py; done
done > gpurun_out/r06/nocs_ab.log 2>&1
python bench.py --no-extras --steps 10 --warmup 3 > gpurun_out/r06/bench0.log 2>&1
python -m pytest tests/test_hip_graph_gpu.py tests/test_grad_homes_gpu.py -x -q -m gpu > gpurun_out/r06/tests0.log 2>&1
tail -3 gpurun_out/r06/tests0.log
cat gpurun_out/r06/nocs_ab.log
tail -c 1500 gpurun_out/r06/bench0.log
