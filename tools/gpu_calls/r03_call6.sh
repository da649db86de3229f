python -m pytest tests/test_lstm_gpu.py -x -q -k "lost_block or pair_rows" 2>&1 | tail -4 > gpurun_out/r03_t6.txt
( time python bench.py --steps 20 --warmup 3 > gpurun_out/r03_bench2.json 2> gpurun_out/r03_bench2.err ) 2>> gpurun_out/r03_t6.txt
tail -12 gpurun_out/r03_bench2.err >> gpurun_out/r03_t6.txt
