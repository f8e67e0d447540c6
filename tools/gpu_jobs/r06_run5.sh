# round 6, GPU session 5: two ranks through the native exchange on one GPU (tests/support/libloopback_ccl.so), the rest of the exchange tests
timeout 900 python -m pytest tests/test_gpu_rccl.py tests/test_gpu_parity.py -x -q -k "rccl or ranks or tile_split or native" 2>&1 | grep -v WARNING | tail -30
