cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3g
( time timeout 900 python -m pytest tests -m gpu -x -q ) > gpurun_out/r3g/gputests.log 2>&1
grep -E "passed|failed" gpurun_out/r3g/gputests.log | tail -2
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
( time bash tools/profile_round.sh r03 f16x3 ) > gpurun_out/r3g/prof_f16x3.log 2>&1
tail -2 gpurun_out/r3g/prof_f16x3.log
