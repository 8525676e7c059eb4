set -x
mkdir -p gpurun_out/r06
python tools/dev/shape_times.py 96 > gpurun_out/r06/shape_times_b96.log 2>&1
grep -v amdgpu.ids gpurun_out/r06/shape_times_b96.log | head -70
python -m pytest tests -q -m gpu --deselect tests/test_large_v3_gpu.py > gpurun_out/r06/tests8_full.log 2>&1
grep -E "passed|failed|FAILED" gpurun_out/r06/tests8_full.log | tail -15
