#!/bin/bash
# Collect the round's rocprofv3 evidence on the GPU box (run through gpurun from the repo root):
#   1. --kernel-trace --stats of the bench command            -> gpurun_out/prof_<tag>/kt_results.db
#   2. --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) -> gpurun_out/prof_<tag>/{fetch,write}_results.db
#   3. --pmc MFMA busy / clock counters                        -> gpurun_out/prof_<tag>/mfma_results.db
# and distil them into text/JSON summaries (copied into profiles/ by hand afterwards).
# PMC passes never combine with sys/runtime/hip traces (gpurun refuses that combination).
set -u
TAG=${1:-r01}
MATH=${2:-}          # optional arithmetic of the profiled step: f16x3 (library default) or f32
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
# (a) the training step alone: every launch of a GEMM kernel in this command has one of the step's 10 shapes, so the
#     per-kernel average duration is directly comparable with bench.py's `kernels.*.avg_launch_ms`
BENCH="python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-rocm-ref --no-eval --no-f32 ${MATH:+--math $MATH}"
rocprofv3 --kernel-trace --stats -d "$OUT" -o kt -- $BENCH > "$OUT/kt.log" 2>&1
{ echo "# rocprofv3 --kernel-trace --stats -- $BENCH"; python "$R/tools/prof_summary.py" "$OUT/kt_results.db" 40; } \
    > "$OUT/${TAG}_bench_train_kernel_trace_stats.txt" 2>&1
# (b) the full default command (adds the cfg2 eval-forward section: the 11 ms k_rows_gemm<true,true> launches)
FULL="python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-rocm-ref --no-f32 ${MATH:+--math $MATH}"
rocprofv3 --kernel-trace --stats -d "$OUT" -o ktfull -- $FULL > "$OUT/ktfull.log" 2>&1
{ echo "# rocprofv3 --kernel-trace --stats -- $FULL"; python "$R/tools/prof_summary.py" "$OUT/ktfull_results.db" 40; } \
    > "$OUT/${TAG}_bench_full_kernel_trace_stats.txt" 2>&1
SHORT="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-rocm-ref --no-eval --no-f32 ${MATH:+--math $MATH}"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$OUT" -o fetch -- $SHORT > "$OUT/fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d "$OUT" -o write -- $SHORT > "$OUT/write.log" 2>&1
python "$R/tools/pmc_traffic.py" "$OUT/fetch_results.db" "$OUT/write_results.db" "$OUT/${TAG}_pmc_traffic.json" \
       "$OUT/${TAG}_pmc_traffic.txt" > "$OUT/pmc_traffic.log" 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY \
          --kernel-trace -d "$OUT" -o mfma -- $SHORT > "$OUT/mfma.log" 2>&1
python "$R/tools/pmc_mfma.py" "$OUT/mfma_results.db" > "$OUT/${TAG}_pmc_mfma_util.txt" 2>&1
rm -f "$OUT"/*_results.db      # the raw databases (10-20 MB each) would push gpurun_out/ past its copy-back limit
ls -la "$OUT"
tail -3 "$OUT/kt.log"
