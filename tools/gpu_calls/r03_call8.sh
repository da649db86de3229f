python -m pytest tests/test_train_forward_gpu.py -x -q 2>&1 | tail -6 > gpurun_out/r03_t8.txt
python tools/bench_train_forward.py --full-step --graph --precision fp32 --steps 3 >> gpurun_out/r03_t8.txt 2>/dev/null
