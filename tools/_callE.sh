cd $GRAFT_REPO_ROOT
for mr in 16384 8192 2048; do
  echo "min rows $mr"
  VP3D_FWD_SPLIT_MIN_ROWS=$mr python tools/env_ab.py VP3D_FWD_SPLIT 0 1 3 30 2>&1 | grep -v amdgpu
done
