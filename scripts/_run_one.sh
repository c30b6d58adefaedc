mkdir -p gpurun_out/$1
timeout 900 python -m pytest tests/test_gpu_vawgan.py -q 2>&1 | tail -25 > gpurun_out/$1/pytest.txt
python scripts/vawgan_bench.py --frames 16 > gpurun_out/$1/bench16.txt 2>&1
python scripts/vawgan_bench.py --frames 256 --iters 20 > gpurun_out/$1/bench256.txt 2>&1
