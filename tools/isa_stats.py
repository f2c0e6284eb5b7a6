#!/usr/bin/env python3
"""Per-kernel resource table of the built library (no GPU needed):  python tools/isa_stats.py [libvp3d.so] [name-filter]

Splits the .hip_fatbin section into its per-translation-unit bundles, unbundles the gfx950 code objects and prints, per
kernel: VGPRs / AGPRs / SGPRs, LDS bytes, private-segment (scratch) bytes, spilled registers (from the code-object metadata)
and the number of scratch_ / v_mfma / buffer_load..lds / ds_read instructions in the disassembly; scr_loop = the scratch
instructions that lie between the kernel's first and last v_mfma (spills inside the K loop; the others are prologue / epilogue)."""
import os
import re
import subprocess
import sys
import tempfile

B = "/opt/rocm/lib/llvm/bin/"
args = sys.argv[1:]
lib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "videopose3d_amd", "libvp3d.so")
if args and os.path.exists(args[0]):
    lib = args.pop(0)
flt = args[0] if args else ""
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
with tempfile.TemporaryDirectory() as d:
    fat = os.path.join(d, "fat.bin")
    subprocess.check_call([B + "llvm-objcopy", "--dump-section", ".hip_fatbin=" + fat, lib])
    blob = open(fat, "rb").read()
    starts = [m.start() for m in re.finditer(re.escape(MAGIC), blob)]
    rows = []
    for i, s in enumerate(starts):
        part = os.path.join(d, "b%d.bin" % i)
        open(part, "wb").write(blob[s:starts[i + 1] if i + 1 < len(starts) else len(blob)])
        co = os.path.join(d, "b%d.co" % i)
        subprocess.check_call([B + "clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950",
                               "--input=" + part, "--output=" + co])
        notes = subprocess.run([B + "llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
        meta = {}
        for blk in notes.split("  - .agpr_count:")[1:]:
            blk = ".agpr_count:" + blk
            g = lambda k: (re.search(r"\.%s:\s*(\S+)" % k, blk) or [None, "?"])[1]
            meta[g("name")] = dict(vgpr=g("vgpr_count"), agpr=g("agpr_count"), sgpr=g("sgpr_count"), lds=g("group_segment_fixed_size"),
                                   scratch=g("private_segment_fixed_size"), spill=g("vgpr_spill_count"), sspill=g("sgpr_spill_count"))
        dis = subprocess.run([B + "llvm-objdump", "-d", co], capture_output=True, text=True).stdout
        cur, cnt = None, {}
        for line in dis.splitlines():
            m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
            if m:
                cur = m.group(1)
                cnt[cur] = dict(scratch=0, mfma=0, ldsdma=0, ds_read=0, insts=0, first_mfma=-1, last_mfma=-1, scr_at=[])
                continue
            if cur is None or "\t" not in line:
                continue
            c = cnt[cur]
            c["insts"] += 1
            if "scratch_" in line:
                c["scratch"] += 1
                c["scr_at"].append(c["insts"])
            if "v_mfma" in line:
                c["mfma"] += 1
                if c["first_mfma"] < 0:
                    c["first_mfma"] = c["insts"]
                c["last_mfma"] = c["insts"]
            if re.search(r"(buffer|global)_load.* lds", line) or "global_load_lds" in line:
                c["ldsdma"] += 1
            if "ds_read" in line or "ds_load" in line:
                c["ds_read"] += 1
        for name, mt in meta.items():
            rows.append((name, mt, cnt.get(name, {})))
    demangle = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.splitlines()
    print("%-5s %-5s %-5s %-7s %-8s %-6s %-7s %-8s %-6s %-6s %-7s %-7s  %s" % ("vgpr", "agpr", "sgpr", "lds", "scratchB", "vspill", "scr_ins", "scr_loop", "mfma", "ldsdma", "ds_read", "insts", "kernel"))
    tot = 0
    for (name, mt, c), dn in sorted(zip(rows, demangle), key=lambda r: r[1]):
        if flt and flt not in dn:
            continue
        tot += c.get("scratch", 0)
        # scr_loop: scratch instructions BETWEEN the kernel's first and last v_mfma (the K loop); the rest sit in prologue / epilogue
        in_loop = sum(1 for x in c.get("scr_at", []) if c.get("first_mfma", -1) <= x <= c.get("last_mfma", -1))
        print("%-5s %-5s %-5s %-7s %-8s %-6s %-7s %-8s %-6s %-6s %-7s %-7s  %s" % (mt["vgpr"], mt["agpr"], mt["sgpr"], mt["lds"], mt["scratch"], mt["spill"],
                                                                       c.get("scratch", "?"), in_loop, c.get("mfma", "?"), c.get("ldsdma", "?"), c.get("ds_read", "?"), c.get("insts", "?"), dn[:150]))
    print("total scratch instructions: %d" % tot)
