for v in 32 0; do echo "## winoq_chsplit_max=$v"; DCE_TUNE=winoq_chsplit_max=$v python tools/latency_small_batch.py 2>&1 | grep -v amdgpu.ids | head -4 | cut -c1-260; done
python -m pytest tests/test_round3_gpu.py tests/test_round6_gpu.py tests/test_gpu_parity.py -q -x -p no:cacheprovider 2>&1 | tail -3
