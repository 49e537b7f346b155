mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_engine.py tests/test_gpu_routes.py tests/test_gpu_prefill_long.py "tests/test_gpu_parity_fullwidth.py::test_full_width_forward_matches_oracle" -q --timeout=900 --durations=12 -k "not configs2" > gpurun_out/new_tests.log 2>&1; echo "pytest rc=$?"; tail -30 gpurun_out/new_tests.log
( time timeout 1200 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_r06a.json 2> gpurun_out/bench_r06a.err ) 2>&1 | tail -3; echo "bench rc=$?"; tail -c 3000 gpurun_out/bench_r06a.json; tail -5 gpurun_out/bench_r06a.err
