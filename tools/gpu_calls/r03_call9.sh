python -m pytest tests/test_train_forward_gpu.py -x -q 2>&1 | tail -6 > gpurun_out/r03_t9.txt
python tools/bench_train_forward.py --full-step --graph --precision f16x3 --steps 3 >> gpurun_out/r03_t9.txt 2>gpurun_out/r03_t9.err
tail -3 gpurun_out/r03_t9.err >> gpurun_out/r03_t9.txt
python tools/bench_train_forward.py --full-step --graph --precision fp32 --steps 3 >> gpurun_out/r03_t9.txt 2>/dev/null
