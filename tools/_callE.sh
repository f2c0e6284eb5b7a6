cd $GRAFT_REPO_ROOT
python - <<'PY' 2>&1 | grep -v amdgpu
import sys
sys.argv=['x']
sys.path.insert(0,'tools')
import s16_tune as T
B=1024
for tag,m,n,k in (("L0 fwd",27648,1024,3072),("L0 dgrad",27648,3072,1024),("L1 fwd/dgrad",27648,1024,1024),("L2 fwd",9216,1024,3072),("L2 dgrad",9216,3072,1024),("L3 fwd/dgrad",9216,1024,1024),("L4 fwd",3072,1024,3072),("L4 dgrad",3072,3072,1024)):
    T.sweep(tag,m,n,k,False)
PY
