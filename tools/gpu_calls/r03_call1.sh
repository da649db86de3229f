python -m pytest tests/test_kernels_gpu.py -x -q -k "h2 or H2" 2>&1 | tail -15 > gpurun_out/r03_t1.txt
python -m pytest tests/test_parity_gpu.py -x -q 2>&1 | tail -15 >> gpurun_out/r03_t1.txt
python bench.py --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/r03_bench1.json 2> gpurun_out/r03_bench1.err
tail -3 gpurun_out/r03_bench1.err >> gpurun_out/r03_t1.txt
