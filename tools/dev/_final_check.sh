python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu | tail -2
python -m pytest tests/test_model_gpu.py tests/test_gelu_aux8_gpu.py tests/test_attn_prescale_gpu.py tests/test_launch_mode_gpu.py tests/test_hip_graph_gpu.py tests/test_grad_homes_gpu.py tests/test_bench_gpu.py -q -m gpu 2>&1 | tail -3
