#!/bin/bash
# The reference's UNMODIFIED run.py, end to end on the MI355X, twice: with `from common.model import *` resolved to this
# package through the import shim (PYTHONPATH) and with the reference's own classes on PyTorch-ROCm.
# Same synthetic Human3.6M-shaped dataset (tools/make_synth_h36m.py), same command lines, same seeds.
#
# The reference checkout is not part of this repository and /root/reference does not exist on the GPU box: the caller stages it
# as a git-ignored tarball (tools/runpy_stage.sh -> _ref_stage.tgz, deleted after the call); it is unpacked OUTSIDE the
# repository (/tmp) and only read there.
#
#   gpurun -- 'bash tools/runpy_e2e.sh [phases]'   (sup semi supdrop eval steps optc long flags prof; default all but long / flags)  -> gpurun_out/runpy/*.log   (tools/runpy_summary.py turns them into profiles/r05_runpy_*)
set -u
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/runpy
mkdir -p "$OUT"
W=/tmp/vp3d_ref
rm -rf $W && mkdir -p $W && tar xzf "$REPO/_ref_stage.tgz" -C $W || { echo "no staged reference"; exit 1; }
cd $W/VideoPose3D || exit 1
sha256sum run.py common/model.py > "$OUT/reference_sha256.txt"
python "$REPO/tools/make_synth_h36m.py" --reference . > "$OUT/make_synth.log" 2>&1 || { cat "$OUT/make_synth.log"; exit 1; }
export MIOPEN_USER_DB_PATH=/tmp/miopen_db MIOPEN_LOG_LEVEL=1
# run.py never seeds torch (its parameter init differs from process to process), so loss TRAJECTORIES of two executions only
# compare when the seed comes from outside: a `usercustomize` module on PYTHONPATH (imported by `site` at interpreter start-up,
# before run.py's first line) calls torch.manual_seed.  Only the runs that say SEEDHOOK=1 use it.
mkdir -p $W/seedhook && printf 'import torch\ntorch.manual_seed(1234)\n' > $W/seedhook/usercustomize.py

one() {   # one <who: ours|ref|refcpu> <name> <run.py args...>
    who=$1; name=$2; shift 2
    log="$OUT/${name}_${who}.log"
    echo "# python run.py $*   [$who]" > "$log"
    t0=$(date +%s.%N)
    hook=""; [ "${SEEDHOOK:-0}" = 1 ] && hook=":$W/seedhook"
    if [ "$who" = ours ]; then
        # -X importtime: which file `common.model` came from is on record (stderr), nothing else changes
        PYTHONPATH="$REPO/videopose3d_amd:$REPO$hook" timeout 1500 python -X importtime run.py "$@" >> "$log" 2> "$OUT/${name}_${who}.err"
    elif [ "$who" = refcpu ]; then      # the reference classes on the host's cores (no device visible to torch)
        PYTHONPATH="${hook#:}" CUDA_VISIBLE_DEVICES=-1 HIP_VISIBLE_DEVICES=-1 ROCR_VISIBLE_DEVICES=-1 timeout 1500 python -X importtime run.py "$@" >> "$log" 2> "$OUT/${name}_${who}.err"
    else
        PYTHONPATH="${hook#:}" timeout 1500 python -X importtime run.py "$@" >> "$log" 2> "$OUT/${name}_${who}.err"
    fi
    rc=$?
    t1=$(date +%s.%N)
    echo "# exit $rc wall_s $(python -c "print('%.1f' % ($t1 - $t0))")" >> "$log"
    grep -E "videopose3d_amd\.model|videopose3d_amd\._lib| common\.model" "$OUT/${name}_${who}.err" | sed 's/^/# import: /' >> "$log"
    grep -v "^import time:" "$OUT/${name}_${who}.err" | tail -20 > "$OUT/${name}_${who}.stderr_tail"
    rm -f "$OUT/${name}_${who}.err"
    tail -4 "$log"
}

PHASES=${*:-sup semi supdrop eval steps optc prof}
has() { case " $PHASES " in *" $1 "*) return 0;; esac; return 1; }
SUP="-k synth -arc 3,3,3,3,3 -e 3 -b 1024 -drop 0 --checkpoint-frequency 1"
SEMI="-k synth -arc 3,3,3 -e 3 -b 1024 -drop 0 -str S1 -sun S5,S6,S7,S8 --warmup 1 --checkpoint-frequency 1"
if has sup; then for who in ours ref; do
    one $who sup $SUP -c ck_sup_$who                                  # supervised, cfg3 shape (run.py:398-420), dropout 0: comparable losses
done; fi
if has semi; then for who in ours ref; do
    one $who semi $SEMI -c ck_semi_$who                               # semi-supervised (run.py:322-396): two models + project_to_2d
done; fi
# the default command line (dropout 0.25; masks differ between the two, so only rates and loss levels compare)
if has supdrop; then for who in ours ref; do
    one $who supdrop -k synth -arc 3,3,3,3,3 -e 2 -b 1024 --checkpoint-frequency 10 -c ck_supdrop_$who
done; fi
# --evaluate: each implementation evaluates BOTH implementations' checkpoints (state_dict interchange in both directions,
# run.py:203-219, 652-721).  Two facts about the reference's own script on a current torch, the same for either implementation:
# (1) run.py:207 calls torch.load() without weights_only=False on a checkpoint that holds a numpy RandomState (run.py:600-608):
# torch >= 2.6 refuses; TORCH_FORCE_NO_WEIGHTS_ONLY_LOAD=1 is torch's own switch for such scripts.  (2) run.py:213 tests
# `'model_traj' in checkpoint`, and a SUPERVISED checkpoint written by run.py:600-608 holds that key with value None, so run.py:219
# raises for its own checkpoints (seen in gpurun_out/c62).  Hence: (a) the semi-supervised checkpoints (both models present) are
# evaluated as they are; (b) the supervised ones in the published format (pretrained_h36m_cpn.bin: no 'model_traj' key).
if has eval; then
    export TORCH_FORCE_NO_WEIGHTS_ONLY_LOAD=1 MIOPEN_FIND_MODE=FAST
    for who in ours ref; do
        [ -f ck_sup_$who/epoch_1.bin ] || one $who evalprep -k synth -arc 3,3,3,3,3 -e 1 -b 1024 -drop 0 --checkpoint-frequency 1 -c ck_sup_$who
        [ -f ck_semi_$who/epoch_1.bin ] || one $who evalprep_semi -k synth -arc 3,3,3 -e 1 -b 1024 -drop 0 -str S1 -sun S5,S6,S7,S8 --warmup 1 --checkpoint-frequency 1 -c ck_semi_$who
        python -c "import torch; c = torch.load('ck_sup_$who/epoch_1.bin', weights_only=False); torch.save({k: c[k] for k in ('epoch', 'lr', 'model_pos')}, 'ck_sup_$who/published_format.bin')"
    done
    for who in ours ref; do
        for ck in ours ref; do
            one $who eval_ck${ck} -k synth -arc 3,3,3,3,3 -c ck_sup_$ck --evaluate published_format.bin
            one $who evalsemi_ck${ck} -k synth -arc 3,3,3 -c ck_semi_$ck --evaluate epoch_1.bin
        done
    done
    unset TORCH_FORCE_NO_WEIGHTS_ONLY_LOAD MIOPEN_FIND_MODE
fi
# per-STEP loss trajectories out of the unmodified script: one batch per epoch (--subset 0.008: 1,056 training windows, -b 1200), so
# the `3d_train` figure run.py prints per epoch is the loss of ONE optimizer step; 12 steps with the lr / BatchNorm-momentum
# schedules of run.py:583-593.  Three executions of the same command: this package, the reference classes on PyTorch-ROCm, the
# reference classes on the CPU (the distance between the last two is the reference's own implementation-to-implementation noise;
# RUNPY_CPU=1, or run that leg on any host: the data and the seed are deterministic).  Same torch seed for all three: see SEEDHOOK above.
if has steps; then
    STEPS="-k synth -arc 3,3,3,3,3 -e 12 -b 1200 --subset 0.008 -drop 0 --no-eval --checkpoint-frequency 100"
    export SEEDHOOK=1 MIOPEN_FIND_MODE=FAST
    one ours steps $STEPS -c ck_steps_ours
    one ref steps $STEPS -c ck_steps_ref
    [ "${RUNPY_CPU:-0}" = 1 ] && one refcpu steps $STEPS -c ck_steps_refcpu      # (also runs anywhere without a GPU: same data, same seed)
    unset SEEDHOOK MIOPEN_FIND_MODE
fi
# what an epoch costs without run.py's evaluation passes, unmodified vs with the opt-in edits of INTEGRATION.md 3b (device
# generators + fused loss + fused Adam: tools/runpy_optc_patch.py writes run_optc.py next to run.py)
if has optc; then
    python "$REPO/tools/runpy_optc_patch.py" . > "$OUT/optc_patch.log" 2>&1
    NOEV="-k synth -arc 3,3,3,3,3 -e 3 -b 1024 --no-eval --checkpoint-frequency 100"
    one ours noeval $NOEV -c ck_ne_ours
    one ref noeval $NOEV -c ck_ne_ref
    cp run.py run_unmodified.py && cp run_optc.py run.py
    one ours optc_noeval $NOEV -c ck_ne_optc
    one ours optc_sup $SUP -c ck_sup_optc
    cp run_unmodified.py run.py
fi
# longer seeded pair: 8 epochs (1,032 optimizer steps) from identical weights, dropout 0 -- how far two implementations drift in
# situ -- and 20 epochs of the opt-in path with the default dropout (what the fast path converges to on this data)
if has long; then
    export SEEDHOOK=1 MIOPEN_FIND_MODE=FAST       # (FAST: no kernel search, but immediate-mode kernels -- the reference ran 103 s per epoch)
    LONG="-k synth -arc 3,3,3,3,3 -e 8 -b 1024 -drop 0 --checkpoint-frequency 100"
    one ours long $LONG -c ck_long_ours
    one ref long $LONG -c ck_long_ref
    python "$REPO/tools/runpy_optc_patch.py" . > "$OUT/optc_patch.log" 2>&1
    cp run.py run_unmodified.py && cp run_optc.py run.py
    one ours optc_long -k synth -arc 3,3,3,3,3 -e 20 -b 1024 --checkpoint-frequency 100 -c ck_long_optc
    cp run_unmodified.py run.py
    unset SEEDHOOK MIOPEN_FIND_MODE
fi
# run.py's other model-facing switches (run.py:171-184, arguments.py:46-59), two short seeded epochs each (--subset 0.1), both
# implementations: causal convolutions, dense (non-dilated) convolutions, the un-optimised dilated class for training, chunked
# training (--stride 9: TemporalModel with 9 output frames per window), no augmentation / no TTA, 17 -> 512 channels
if has flags; then
    # (no MIOPEN_FIND_MODE=FAST here: it skips the ~85-s kernel search per process but then runs the reference's convolutions on
    #  MIOpen's immediate-mode kernels -- measured 103 s per epoch instead of 10 in the `long` phase)
    export SEEDHOOK=1
    FL="-k synth -arc 3,3,3 -e 2 -b 1024 -drop 0 --subset 0.1 --checkpoint-frequency 100"
    one ours flag_causal $FL --causal -c ck_f1_ours
    one ours flag_dense $FL --dense -c ck_f2_ours
    one ours flag_noopt $FL --disable-optimizations -c ck_f3_ours
    one ours flag_stride9 $FL --stride 9 -c ck_f4_ours
    one ours flag_noaug $FL -no-da -no-tta -c ck_f5_ours
    one ours flag_ch512 $FL -ch 512 -c ck_f6_ours
    one ref flag_causal $FL --causal -c ck_f1_ref
    one ref flag_stride9 $FL --stride 9 -c ck_f4_ref
    unset SEEDHOOK
fi
# kernel-level evidence that the shim run executes this package's HIP kernels: one short epoch under rocprofv3
if has prof; then
cd /tmp && export TMPDIR=/tmp
( cd $W/VideoPose3D && PYTHONPATH="$REPO/videopose3d_amd:$REPO" timeout 900 rocprofv3 --kernel-trace --stats -d "$OUT/prof" -o rp -- \
    python run.py -k synth -arc 3,3,3,3,3 -e 1 -b 1024 --subset 0.3 --no-eval -c ck_prof > "$OUT/prof_run.log" 2>&1 )
{ echo "# rocprofv3 --kernel-trace --stats -- python run.py -k synth -arc 3,3,3,3,3 -e 1 -b 1024 --subset 0.3 --no-eval   [ours, import shim]";
  python "$REPO/tools/prof_summary.py" "$OUT/prof/rp_results.db" 30; } > "$OUT/runpy_kernel_trace_stats.txt" 2>&1
rm -rf "$OUT/prof"
fi
ls -la "$OUT"
