set -x
mkdir -p gpurun_out/r06
bash profiles/steady_state_launches.sh r06_ss > gpurun_out/r06/steady_state.log 2>&1
grep -v amdgpu.ids gpurun_out/r06/steady_state.log | tail -45
python -m pytest tests/test_large_v3_gpu.py -q -m gpu -s > gpurun_out/r06/tests11_large.log 2>&1
grep -E "passed|failed|FAILED|largest error outside" gpurun_out/r06/tests11_large.log | cut -c1-300
