#!/bin/bash
# kernel-trace timeline of one training step of the bench configuration -> gpurun_out/timeline.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/tl
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $OUT -o tl -- python $R/tools/s16_prof.py f16x3 train 6 > $OUT/tl.log 2>&1
python $R/tools/timeline.py $OUT/tl_results.db > $R/gpurun_out/timeline.txt 2>&1
rm -f $OUT/*.db
tail -3 $R/gpurun_out/timeline.txt
