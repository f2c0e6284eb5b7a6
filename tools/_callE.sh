cd $GRAFT_REPO_ROOT
for n in 0 4 8 2; do
  echo "VP3D_SIDE_CU_SKIP=$n"
  VP3D_SIDE_CU_SKIP=$n python tools/env_ab.py VP3D_OVERLAP 1 1 2 25 2>&1 | grep -v amdgpu | tail -1
done
