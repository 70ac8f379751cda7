cd /root/repo
timeout 600 python -m pytest tests/test_gemm_gpu.py tests/test_race_screen_gpu.py -x -q -m gpu -k "tiled or gate_up or race or repeated or full_size or config3" 2>&1 | tail -2
echo "== OLD"; (cd _ab_old && timeout 200 python scripts/bench_gemm_big.py 4096x4096x4096 65536x28672x4096 2>&1 | grep -v amdgpu.ids)
echo "== NEW"; timeout 200 python scripts/bench_gemm_big.py 4096x4096x4096 65536x28672x4096 2>&1 | grep -v amdgpu.ids
