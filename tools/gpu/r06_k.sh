python -m pytest tests -m gpu -x -q -rP 2>&1 > gpurun_out/r06_k_gpu_tests_full.log; tail -3 gpurun_out/r06_k_gpu_tests_full.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06_k_smoke.log 2>&1; tail -1 gpurun_out/r06_k_smoke.log
python bench.py > gpurun_out/r06_k_bench.json 2> gpurun_out/r06_k_bench.err; tail -c 400 gpurun_out/r06_k_bench.json
