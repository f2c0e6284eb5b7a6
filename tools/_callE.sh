cd $GRAFT_REPO_ROOT
( timeout 300 python -m pytest tests/test_gpu_s16.py tests/test_gpu_tail.py -m gpu -x -q -k "fused_prologue or tail" ) 2>&1 | tail -5
