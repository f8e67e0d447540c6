# round 6, GPU session 13: RT_FETCH_BLOCK_MAX 128 (shipped) / 256 / 384 on other launch sizes: the run's frames one after the other (--burst 0), rank 0's share of an
# 8-GPU split (--emulate-world 8), the default 64-step command (bursts of eight frames), config 3 (SVGF + TAA, one sample per frame)
mkdir -p gpurun_out
V=gpu-raytracer_amd/csrc/_variants
for name in shipped fb256 fb384; do
  lib=""; case $name in shipped) ;; *) lib="$PWD/$V/$name/libgrt_device.so";; esac
  for work in "burst0:--steps 20 --warmup 5 --burst 0 --no-config3" "world8:--steps 20 --warmup 5 --emulate-world 8 --no-config3" "steps64:--steps 64 --warmup 4 --no-config3" "config3:--steps 20 --warmup 5 --no-stages"; do
    tag=${work%%:*}; args=${work#*:}
    GRT_DEVICE_LIB=$lib timeout 400 python bench.py --gpus 1 $args --no-cpu-baseline --no-povs --no-pmc --no-reference-layout > gpurun_out/r06_run13_${name}_$tag.json 2> gpurun_out/r06_run13_${name}_$tag.err
    python - <<PY
import json
try:
    d = json.load(open("gpurun_out/r06_run13_${name}_$tag.json")); r = d["roofline"]; st = {s["stage"]: s["ms_per_step"] for s in r.get("stages", [])}; c3 = d.get("config3") or {}
    print("%-8s %-8s %.4f ms/step  traversal %.4f | config3 %s ms per frame (traversal %s)" % ("$name", "$tag", d["ms_per_step"], st.get("traversal", 0), c3.get("ms_per_filtered_frame"), c3.get("traversal_ms_per_frame")))
except Exception as e: print("$name $tag failed", e); print(open("gpurun_out/r06_run13_${name}_$tag.err").read()[-600:])
PY
  done
done
