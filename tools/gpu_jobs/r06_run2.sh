# round 6, GPU session 2: where the skipping walk's time goes -- variants of the flattened scene's launch, the driver's command without the side sections:
#   walk0        the reference's walk (config skip_behind_hit 0)
#   skip         the shipped skipping walk (two-entry peek at a pop)
#   nodrop       bounds computed and carried, nothing ever dropped (the price of the bound alone)
#   looppop      pops until a group in front of the hit is found (a loop of LDS round trips)
#   skip5/skip7  the skipping launch compiled for 5 / 7 waves per SIMD
mkdir -p gpurun_out
B="--no-cpu-baseline --no-povs --no-pmc --no-config3 --no-reference-layout"
V=gpu-raytracer_amd/csrc/_variants
for name in walk0 skip nodrop looppop skip5 skip7 walk0_b skip_b; do
  lib=""; v=1
  case $name in walk0*) v=0;; skip|skip_b) ;; *) lib="$PWD/$V/$name/libgrt_device.so";; esac
  case $name in looppop|skip5|skip7) GRT_DEVICE_LIB=$lib timeout 300 python -m pytest tests/test_gpu_static_geometry.py tests/test_gpu_parity.py -x -q -k "bit_exact or flattened or statistics" 2>&1 | tail -1;; esac
  GRT_DEVICE_LIB=$lib BENCH_SKIP_BEHIND_HIT=$v timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 $B > gpurun_out/r06_run2_$name.json 2> gpurun_out/r06_run2_$name.err
  python - <<PY
import json
try:
    d = json.load(open("gpurun_out/r06_run2_$name.json")); r = d["roofline"]; st = {s["stage"]: s["ms_per_step"] for s in r.get("stages", [])}
    print("%-10s %.4f ms/step  traversal %.4f | nodes/tris %s %s" % ("$name", d["ms_per_step"], st.get("traversal", 0), r.get("nodes_per_ray"), r.get("triangles_per_ray")))
except Exception as e: print("$name failed", e); print(open("gpurun_out/r06_run2_$name.err").read()[-1500:])
PY
done
