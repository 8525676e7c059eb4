set -x
mkdir -p gpurun_out/r06
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06/smoke13.log 2>&1
tail -3 gpurun_out/r06/smoke13.log
python -m pytest tests -q -m gpu > gpurun_out/r06/tests13_full.log 2>&1
grep -E "passed|failed|FAILED" gpurun_out/r06/tests13_full.log | tail -10
