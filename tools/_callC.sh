cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3c
( time timeout 900 python -m pytest tests -m gpu -x -q ) > gpurun_out/r3c/gputests.log 2>&1
tail -4 gpurun_out/r3c/gputests.log
( time bash tools/profile_round.sh r03 f16x3 ) > gpurun_out/r3c/prof_f16x3.log 2>&1
( time bash tools/profile_round.sh r03f32 f32 ) > gpurun_out/r3c/prof_f32.log 2>&1
tail -3 gpurun_out/r3c/prof_f16x3.log gpurun_out/r3c/prof_f32.log
