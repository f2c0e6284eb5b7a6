cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3f
( time python bench.py ) > gpurun_out/r3f/bench_full.json 2> gpurun_out/r3f/bench_full.err
tail -3 gpurun_out/r3f/bench_full.err
