# round 6, GPU session 34: long runs (160 steps = 40 frames in bursts of eight) on the round's last state: whole frame and rank 0 of 2 / 4 / 8; 64 steps (the default of `python bench.py`)
mkdir -p gpurun_out
B="--no-cpu-baseline --no-povs --no-pmc --no-config3 --no-reference-layout --no-stages --no-tile-split-bound"
for S in 64 160; do for W in 0 2 4 8; do
  timeout 300 python bench.py --gpus 1 --steps $S --warmup 8 --emulate-world $W $B 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('steps $S world $W: %.4f ms/step  %.1f Mrays/s' % (d['ms_per_step'], d['value']))"
done; done
