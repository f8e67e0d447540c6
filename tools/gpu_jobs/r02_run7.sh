mkdir -p gpurun_out
python - <<'PY' 2>&1 | grep -v WARNING
import sys, time, os; sys.path.insert(0,'.'); sys.path.insert(0,'tools')
import numpy as np
import gpu_raytracer_amd as grt, bench
from oracle import binding as oracle
scene=bench.build_scene(grt); pt=grt.Pathtracer(scene,1920,1080,device=-1); pt.update()
view=oracle.SceneView(pt)
o,d,_=view.generate(0,0,1920*1080)
hits,_=view.trace(o,d,64)
t=hits[:,2].view(np.float32); ok=hits[:,1]!=0xffffffff
rng=np.random.default_rng(1); so=(o+d*np.where(ok,t,1).astype(np.float32)*np.float32(0.999))[:,ok]; sd=rng.normal(size=so.shape).astype(np.float32); sd/=np.linalg.norm(sd,axis=0)
for th in (8,16,32,64,128,256):
    t0=time.perf_counter(); view.trace(so,sd,th); dt=time.perf_counter()-t0
    print("oracle secondary rays: %3d threads %.2f s  %.2f Mrays/s" % (th, dt, so.shape[1]/dt/1e6), flush=True)
PY
(time timeout 600 python -m pytest tests/test_gpu_tlas.py tests/test_gpu_parity.py -x -q -k "animated or moving" ) > gpurun_out/r02_run7_tests.log 2>&1; echo "tests rc=$?"; tail -4 gpurun_out/r02_run7_tests.log
timeout 300 python tools/animation_bench.py > gpurun_out/r02_animation_441.log 2>&1; grep TLAS gpurun_out/r02_animation_441.log
ANIM_INSTANCES=1500 timeout 300 python tools/animation_bench.py > gpurun_out/r02_animation_1500.log 2>&1; grep TLAS gpurun_out/r02_animation_1500.log
ANIM_INSTANCES=4000 timeout 300 python tools/animation_bench.py > gpurun_out/r02_animation_4000.log 2>&1; grep TLAS gpurun_out/r02_animation_4000.log
