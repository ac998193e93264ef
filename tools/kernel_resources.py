#!/usr/bin/env python3
"""Registers / scratch / LDS / occupancy of every kernel of the product library, from hipcc's kernel-resource-usage remarks
(cross-compiles for gfx950: no GPU needed).  usage: tools/kernel_resources.py [extra hipcc flags ...] [-- name-filter]"""
import os, re, subprocess, sys, tempfile
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
filt = ""
if "--" in args:
    i = args.index("--"); filt = " ".join(args[i + 1:]); args = args[:i]
with tempfile.TemporaryDirectory() as d:
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-Wno-unused-value",
           "-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(root, "liodom_amd/csrc/liodom_hip.hip"), "-o", os.path.join(d, "x.o")] + args
    txt = subprocess.run(cmd, capture_output=True, text=True).stderr
for b in re.split(r"remark: [^\n]*Function Name: ", txt)[1:]:
    mangled = b.split()[0]
    try:
        name = subprocess.run(["c++filt", mangled], capture_output=True, text=True).stdout.strip().split("(")[0]
    except Exception:
        name = mangled
    name = name.replace("void liodom_dev::", "").replace("liodom_dev::", "")
    if filt and filt not in name:
        continue
    def g(k):
        m = re.search(k + r": (\d+)", b)
        return m.group(1) if m else "?"
    print("%-40s vgpr %4s sgpr %4s scratch %5s occ %2s lds %6s" % (name[:40], g("VGPRs"), g("SGPRs"), g(r"ScratchSize \[bytes/lane\]"),
                                                                   g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]")))
