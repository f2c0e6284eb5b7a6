set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3a
( time python -m pytest tests -m gpu -x -q -s -k "headline_workload or full_gradients or training_loop_vs" ) > gpurun_out/r3a/newtests.log 2>&1
( time python -m pytest tests -m gpu -x -q ) > gpurun_out/r3a/gputests.log 2>&1
python tools/smi_interference.py 3 0.5 > gpurun_out/r3a/smi.log 2>&1
python tools/graph_branches.py gpurun_out/r3a/graph_default.dot > gpurun_out/r3a/graph_default.log 2>&1
DEBUG_HIP_FORCE_GRAPH_QUEUES=8 python tools/graph_branches.py > gpurun_out/r3a/graph_q8.log 2>&1
DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 python tools/graph_branches.py > gpurun_out/r3a/graph_nopkt.log 2>&1
VP3D_OVERLAP=0 python tools/graph_branches.py > gpurun_out/r3a/graph_noov.log 2>&1
python bench.py --no-rocm-ref --no-eval --no-f32 > gpurun_out/r3a/bench.json 2> gpurun_out/r3a/bench.err
tail -c 600 gpurun_out/r3a/newtests.log; tail -3 gpurun_out/r3a/gputests.log; cat gpurun_out/r3a/graph_*.log | grep -v Warning | tail -20
