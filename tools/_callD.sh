cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3d
( cd tools/ubench && ./kloop ) > gpurun_out/r3d/kloop.txt 2>&1
( time timeout 900 python -m pytest tests -m gpu -x -q ) > gpurun_out/r3d/gputests.log 2>&1
grep -E "passed|failed" gpurun_out/r3d/gputests.log | tail -2
VP3D_TAIL=1 python tools/tail_trace.py > gpurun_out/r3d/tail_trace.txt 2>&1
python tools/tail_ab.py > gpurun_out/r3d/tail_ab.txt 2>&1
cat gpurun_out/r3d/tail_ab.txt | grep -v amdgpu
