cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | grep -v WARNING | tail -3
timeout 300 python bench.py --steps 32 --warmup 4 --no-cpu-baseline 2>&1 | grep -v WARNING | tail -1
for v in t1w5 t3w5 t3w4 t2w4 t2w6 t4w4; do
  echo "=== $v"
  export GRT_DEVICE_LIB=$GRAFT_REPO_ROOT/gpu-raytracer_amd/csrc/_variants/libgrt_device_$v.so
  timeout 120 python tools/trace_bench.py 20000 777600 2073600 2>&1 | grep -v WARNING
done
