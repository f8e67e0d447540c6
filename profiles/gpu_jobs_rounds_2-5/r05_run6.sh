# round 5, sixth GPU session: does a second (third, fourth) wavefront on the same GPU hide the tails of a rank's launches? (tools/dual_wavefront_probe.py)
mkdir -p gpurun_out
( time timeout 600 python tools/dual_wavefront_probe.py --steps 20 --configs N1x1,N1x2,N8x1,N8x2,N8x3,N8x4,N4x1,N4x2,N2x1,N2x2 2>&1 | grep -v WARNING | tee gpurun_out/r05_dual_wavefront.txt ) 2>&1 | tail -14
timeout 300 python tools/dual_wavefront_probe.py --steps 160 --configs N8x1,N8x2,N8x3 --repeat 3 2>&1 | grep -v WARNING | tee -a gpurun_out/r05_dual_wavefront.txt | tail -4
cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/r05_prof6 -o probe -- python $GRAFT_REPO_ROOT/tools/dual_wavefront_probe.py --steps 20 --configs N8x2 --repeat 1 > /dev/null 2>&1; cd $GRAFT_REPO_ROOT
DB=$(find gpurun_out/r05_prof6 -name "*.db" | head -1)
python tools/rocpd_timeline.py $DB 7 > gpurun_out/r05_dual_wavefront_timeline.txt 2>&1; wc -l gpurun_out/r05_dual_wavefront_timeline.txt
rm -rf gpurun_out/r05_prof6
