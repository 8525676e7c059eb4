set -x
mkdir -p gpurun_out/r06
python -m pytest tests/test_gelu_aux8_gpu.py tests/test_attn_fwd_pipe_gpu.py -q -m gpu > gpurun_out/r06/tests9a.log 2>&1
tail -4 gpurun_out/r06/tests9a.log
python bench.py > gpurun_out/r06/bench_full_a.log 2> gpurun_out/r06/bench_full_a.err
tail -1 gpurun_out/r06/bench_full_a.log | cut -c1-600
tail -5 gpurun_out/r06/bench_full_a.err
python -m pytest tests/test_large_v3_gpu.py -q -m gpu -s > gpurun_out/r06/tests9_large.log 2>&1
grep -E "passed|failed|FAILED|^\[" gpurun_out/r06/tests9_large.log | cut -c1-400 | tail -40
