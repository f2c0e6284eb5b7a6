#!/bin/bash
# Collect the round's rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   1. --kernel-trace --stats of the bench command                 -> <tag>_bench_*_kernel_trace_stats.txt
#   2. per-launch table of the training step (tools/step_table.py): kernel-trace + --pmc FETCH_SIZE / WRITE_SIZE (separate
#      passes) + MFMA busy / clocks, joined with the host's launch log -> <tag>_step_table.txt
#   3. per-kernel PMC summaries of the same passes                  -> <tag>_pmc_traffic.{json,txt}, <tag>_pmc_mfma_util.txt
#   4. two-stream timeline of the real step                         -> <tag>_step_timeline.txt
# PMC passes never combine with sys/runtime/hip traces (gpurun refuses that combination).  Summaries are copied into
# profiles/ by hand afterwards.
set -u
TAG=${1:-r02}
MATH=${2:-f16x3}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-rocm-ref --no-eval --no-f32 --no-sweep --math $MATH"
VP3D_BENCH_GRAPH=0 rocprofv3 --kernel-trace --stats -d "$OUT" -o kt -- $BENCH > "$OUT/kt.log" 2>&1
{ echo "# rocprofv3 --kernel-trace --stats -- $BENCH"; python "$R/tools/prof_summary.py" "$OUT/kt_results.db" 40; } \
    > "$OUT/${TAG}_bench_train_kernel_trace_stats.txt" 2>&1
FULL="python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-rocm-ref --no-f32 --no-sweep --math $MATH"
VP3D_BENCH_GRAPH=0 rocprofv3 --kernel-trace --stats -d "$OUT" -o ktfull -- $FULL > "$OUT/ktfull.log" 2>&1
{ echo "# rocprofv3 --kernel-trace --stats -- $FULL"; python "$R/tools/prof_summary.py" "$OUT/ktfull_results.db" 40; } \
    > "$OUT/${TAG}_bench_full_kernel_trace_stats.txt" 2>&1
STEP="python $R/tools/step_table.py run $OUT/hostlog.json $MATH"
rocprofv3 --kernel-trace -d "$OUT" -o st -- $STEP > "$OUT/st.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$OUT" -o fetch -- $STEP > "$OUT/fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d "$OUT" -o write -- $STEP > "$OUT/write.log" 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY \
          --kernel-trace -d "$OUT" -o mfma -- $STEP > "$OUT/mfma.log" 2>&1
python "$R/tools/step_table.py" join "$OUT/hostlog.json" "$OUT/st_results.db" "$OUT/fetch_results.db" "$OUT/write_results.db" \
       "$OUT/mfma_results.db" "$OUT/${TAG}_step_table.json" > "$OUT/${TAG}_step_table.txt" 2>&1
python "$R/tools/pmc_traffic.py" "$OUT/fetch_results.db" "$OUT/write_results.db" "$OUT/${TAG}_pmc_traffic.json" \
       "$OUT/${TAG}_pmc_traffic.txt" > "$OUT/pmc_traffic.log" 2>&1
python "$R/tools/pmc_mfma.py" "$OUT/mfma_results.db" > "$OUT/${TAG}_pmc_mfma_util.txt" 2>&1
rocprofv3 --kernel-trace -d "$OUT" -o tl -- python $R/tools/s16_prof.py $MATH train 6 > "$OUT/tl.log" 2>&1
python "$R/tools/timeline.py" "$OUT/tl_results.db" > "$OUT/${TAG}_step_timeline.txt" 2>&1
# the raw databases (10-20 MB each) would push gpurun_out/ past its copy-back limit
find "$R/gpurun_out/prof_$TAG" -maxdepth 1 -name '*_results.db' -delete
ls -la "$OUT"
tail -5 "$OUT/${TAG}_step_table.txt"
