cd $GRAFT_REPO_ROOT
python tools/tail_ab.py 2>&1 | grep -v amdgpu.ids
