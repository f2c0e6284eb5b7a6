"""Internal A/B switches -- test and measurement hooks, not configuration.

Every entry selects between two implementations that are BOTH live in the package (the non-default one serves other
configurations: narrower models, synchronised BatchNorm, dilated convs, ...).  The parity suite flips them to hold the default
path bit-for-bit (or to summation order) against its alternative on the same model; the A/B tools under tools/ flip them to time
both.  They used to be VP3D_* environment variables read at call time (rounds 2-4); the settled ones moved here in round 5 so that
the environment surface of the package is what a user may actually want to set (README: VP3D_MATH, VP3D_RANGE_GUARD,
VP3D_OVERLAP, VP3D_S16_MIN_GFLOP, VP3D_FUSE_BN_RED, VP3D_GRAPH_PIECEWISE, VP3D_RCCL_CHANNELS, VP3D_DEV_KERNARG).

    from videopose3d_amd._switches import SW;  monkeypatch.setitem(SW, "wgrad_rows", False)
"""

SW = {
    "wgrad_rows": True,        # C x C weight gradients from the S16 rows (vp3d_wgrad_rows_s16) instead of transposed copies
    "prologue_fused": True,    # the split-fp16 forward's prologue as two launches (vp3d_prologue_a/b_s16) instead of seven
    "prologue_overlap": False, # the C x C weight packs of the fused prologue on the second stream, beside the expand layer's statistics
                               # (round 6): bit-identical and NO gain (4.140 vs 4.149 ms, 6 randomised pairs) -- the pack launch takes every
                               # CU slot, the statistics kernel beside it stretches from 26 to 47-78 us (profiles/r06_prologue_overlap.txt)
    "head_kernels": True,      # the shrink conv + its backward on the dedicated head kernels (csrc/vp3d_head.hip) up to 4096 rows (round 6)
    "expand_kernel": True,     # dedicated expand-layer kernels (one-pass input staging, vp3d_expand_fwd_s16, fused P GEMM)
    "expand_fused": True,      # expand layer: BatchNorm + ReLU + dropout in the GEMM epilogue (its conv output never reaches HBM)
    "act_bits": True,          # backward reads stored activation bits instead of regenerating the Philox masks
    "expand_rows": False,      # expand backward's P = G^T X from the S16 rows (measured slower: tools/expand_bwd_bench.py)
    "tile_mix": "1",           # 224- / 160-row tilings: "0" never, "1" planner decides, "2" / "3" only <= / > 16,384-row launches
    "fin_in_finish": False,    # K-sliced forward launches finalise their BatchNorm statistics in the finishing pass (vp3d_s16_fin):
                               # bit-identical, three launches fewer, and 0.8 % SLOWER (profiles/r05_fin_in_finish_ab.txt): off
    "fuse_act_bwd": "0",       # fp32 engine: activation backward inside the dgrad epilogue ("1" wherever legal, "auto")
}
