#!/usr/bin/env python3
"""gpurun_out/runpy/*.log (tools/runpy_e2e.sh) -> one text summary: per run the command, exit code, wall time, which file
`common.model` resolved to, the per-epoch lines run.py printed and its final protocol errors; then the side-by-side loss tables.

    python tools/runpy_summary.py gpurun_out/runpy > profiles/r05_runpy_summary.txt"""
import glob
import os
import re
import sys


def parse(path):
    r = {"cmd": None, "rc": None, "wall": None, "epochs": [], "proto": {}, "shim": False, "info": []}
    for ln in open(path, errors="replace"):
        ln = ln.rstrip("\n")
        if ln.startswith("# python"):
            r["cmd"] = ln[2:]
        elif ln.startswith("# exit"):
            m = re.match(r"# exit (\d+) wall_s ([\d.]+)", ln)
            r["rc"], r["wall"] = int(m.group(1)), float(m.group(2))
        elif ln.startswith("# import:") and "videopose3d_amd.model" in ln:
            r["shim"] = True
        elif ln.startswith("["):
            m = re.match(r"\[(\d+)\] time ([\d.]+) lr ([\d.]+) (.*)", ln)
            if m:
                kv = m.group(4).split()
                r["epochs"].append((int(m.group(1)), float(m.group(2)), dict(zip(kv[0::2], map(float, kv[1::2])))))
        elif ln.startswith("INFO:"):
            r["info"].append(ln[6:])
        else:
            m = re.match(r"Protocol #(\d).*action-wise average: ([\d.]+) mm", ln)
            if m:
                r["proto"][int(m.group(1))] = float(m.group(2))
            m = re.match(r"Velocity.*action-wise average: ([\d.]+) mm", ln)
            if m:
                r["proto"]["vel"] = float(m.group(1))
    return r


def main():
    d = sys.argv[1]
    runs = {}
    for p in sorted(glob.glob(os.path.join(d, "*.log"))):
        name = os.path.basename(p)[:-4]
        if name in ("make_synth", "prof_run", "optc_patch"):
            continue
        runs[name] = parse(p)
    print("# The reference's run.py on one MI355X, synthetic Human3.6M-shaped data (tools/make_synth_h36m.py): `ours` = run.py UNMODIFIED with")
    print("# `from common.model import *` resolved to this package through PYTHONPATH (import shim); `ref` = the reference's own classes on")
    print("# PyTorch-ROCm (MIOpen); `refcpu` = the reference's classes with no device visible.  Collected by tools/runpy_e2e.sh.")
    for f in ("reference_sha256.txt", "make_synth.log"):
        p = os.path.join(d, f)
        if os.path.exists(p):
            print("# %s: %s" % (f, " | ".join(open(p).read().split("\n")[:3])))
    print()
    for name, r in runs.items():
        print("== %s: %s" % (name, r["cmd"]))
        print("   exit %s, wall %.1f s, common.model -> %s" % (r["rc"], r["wall"] or -1,
                                                              "videopose3d_amd (shim)" if r["shim"] else "the reference's common/model.py"))
        for e, t, kv in r["epochs"]:
            print("   [%d] %.2f min  %s" % (e, t, "  ".join("%s %.4f" % (k, v) for k, v in kv.items())))
        if r["proto"]:
            print("   final evaluation: " + "  ".join("%s %.1f mm" % ("P#%s" % k if k != "vel" else "MPJVE", v) for k, v in r["proto"].items()))
        print()

    def table(title, names, key="3d_train"):
        have = [n for n in names if n in runs and runs[n]["epochs"]]
        if len(have) < 2:
            return
        print("-- %s (%s, mm) --" % (title, key))
        print("   step/epoch  " + "  ".join("%16s" % n for n in have) + "   |ours-ref|/ref   |refcpu-ref|/ref")
        n_ep = min(len(runs[n]["epochs"]) for n in have)
        for i in range(n_ep):
            vals = [runs[n]["epochs"][i][2].get(key, float("nan")) for n in have]
            d1 = abs(vals[0] - vals[1]) / abs(vals[1]) if len(vals) > 1 else float("nan")
            d2 = abs(vals[2] - vals[1]) / abs(vals[1]) if len(vals) > 2 else float("nan")
            print("   %10d  " % (i + 1) + "  ".join("%16.6f" % v for v in vals) + "   %.2e        %.2e" % (d1, d2))
        print()
    table("per-STEP training loss, one batch per epoch (dropout 0)", ["steps_ours", "steps_ref", "steps_refcpu"])
    table("supervised, 129 steps per epoch (dropout 0)", ["sup_ours", "sup_ref"])
    table("supervised, 129 steps per epoch (dropout 0), eval-mode loss on the training set", ["sup_ours", "sup_ref"], "3d_eval")
    table("semi-supervised (dropout 0)", ["semi_ours", "semi_ref"])
    table("default dropout 0.25 (different mask streams)", ["supdrop_ours", "supdrop_ref"])
    table("8 epochs from identical weights (seed hook), dropout 0", ["long_ours", "long_ref"])
    table("8 epochs from identical weights (seed hook), dropout 0, eval-mode loss on the test subjects", ["long_ours", "long_ref"], "3d_valid")
    for flag in ("causal", "dense", "noopt", "stride9", "noaug", "ch512"):
        table("run.py switch `%s`, two seeded epochs on a tenth of the data" % flag, ["flag_%s_ours" % flag, "flag_%s_ref" % flag])
    # the SAME checkpoint evaluated by both implementations: per-action protocol-1 error at run.py's full print precision
    def per_action(path):
        out, act = {}, None
        for ln in open(path, errors="replace"):
            m = re.match(r"----(.+)----", ln.strip())
            if m:
                act = m.group(1)
            m = re.match(r"Protocol #1 Error \(MPJPE\): ([\d.]+) mm", ln.strip())
            if m and act:
                out[act] = float(m.group(1))
        return out
    for title, a, b in (("supervised checkpoint written by this package", "eval_ckours_ours", "eval_ckours_ref"),
                        ("supervised checkpoint written by the reference", "eval_ckref_ours", "eval_ckref_ref"),
                        ("semi-supervised checkpoint written by this package", "evalsemi_ckours_ours", "evalsemi_ckours_ref"),
                        ("semi-supervised checkpoint written by the reference", "evalsemi_ckref_ours", "evalsemi_ckref_ref")):
        pa, pb = os.path.join(d, a + ".log"), os.path.join(d, b + ".log")
        if not (os.path.exists(pa) and os.path.exists(pb)):
            continue
        ea, eb = per_action(pa), per_action(pb)
        if not ea or set(ea) != set(eb):
            continue
        print("-- --evaluate, %s: protocol-1 MPJPE per action (mm), evaluated by ours / by the reference classes --" % title)
        for k in ea:
            print("   %-14s %.10f   %.10f   |diff| %.2e mm" % (k, ea[k], eb[k], abs(ea[k] - eb[k])))
        print("   max |diff| %.2e mm (north_star: within 0.1 mm)" % max(abs(ea[k] - eb[k]) for k in ea))
        print()
    print("-- seconds per epoch (run.py's own `time`, minutes x 60) --")
    for n, r in runs.items():
        if r["epochs"]:
            print("   %-22s %s" % (n, "  ".join("%.1f" % (t * 60) for _, t, _ in r["epochs"])))


if __name__ == "__main__":
    main()
