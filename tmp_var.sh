cd $GRAFT_REPO_ROOT
run() { timeout 300 python bench.py --no-cpu-baseline "$@" 2>&1 | grep -v WARNING | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print(d['value'], d['ms_per_step'], r['frac'], r['avg_launch_ms'])"; }
for f in 1 2 3 4; do echo "w1 in flight $f"; run --samples-in-flight $f; done
for f in 1 2 3 4; do echo "w8 in flight $f"; run --emulate-world 8 --samples-in-flight $f; done
echo "w1 in flight 3 steps 96"; run --samples-in-flight 3 --steps 96
