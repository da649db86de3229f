python -m pytest tests/test_train_forward_gpu.py -x -q 2>&1 | tail -15 > gpurun_out/r03_t5.txt
python -m pytest tests/test_lstm_gpu.py -x -q 2>&1 | tail -8 >> gpurun_out/r03_t5.txt
python -m pytest tests/test_parity_gpu.py -x -q -k "tail_windows" 2>&1 | tail -5 >> gpurun_out/r03_t5.txt
python -m pytest tests/test_kernels_gpu.py -x -q 2>&1 | tail -5 >> gpurun_out/r03_t5.txt
