# the tile split's upper bound with the flattened scene: rank 0's share of an 8-way (and 4-, 2-way) split on one GPU against the whole frame
cd /root/repo
mkdir -p gpurun_out
for spec in "0 20" "8 20" "0 160" "8 160" "4 20" "2 20"; do
  set -- $spec
  timeout 300 python bench.py --gpus 1 --steps $2 --warmup 5 --emulate-world $1 --no-cpu-baseline --no-povs --no-pmc --no-config3 --no-stages > gpurun_out/r03_emulate_w$1_s$2.json 2> gpurun_out/r03_emulate_w$1_s$2.err
  python -c "
import json; d=json.load(open('gpurun_out/r03_emulate_w$1_s$2.json')); print('emulate-world $1, $2 steps: %.4f ms/step' % d['ms_per_step'])"
done
