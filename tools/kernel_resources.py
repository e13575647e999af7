"""Summarise `hipcc -Rpass-analysis=kernel-resource-usage` remarks (stderr of a compile) per kernel instantiation.
usage: python tools/kernel_resources.py remarks.txt [substring filters...]"""
import re
import subprocess
import sys

txt = open(sys.argv[1]).read()
filters = sys.argv[2:]
blocks = re.split(r'(?=remark: [^\n]*Function Name)', txt)
rows = []
for b in blocks:
    m = re.search(r'Function Name: (\S+)', b)
    if not m:
        continue
    d = {}
    for k, pat in (("vgpr", r"\bVGPRs"), ("agpr", r"AGPRs"), ("sgpr", r"\bSGPRs"), ("scratch", r"ScratchSize \[bytes/lane\]"),
                   ("occ", r"Occupancy \[waves/SIMD\]"), ("sgpr_spill", r"SGPRs Spill"), ("vgpr_spill", r"VGPRs Spill"),
                   ("lds", r"LDS Size \[bytes/block\]")):
        mm = re.search(pat + r": (\d+)", b)
        d[k] = int(mm.group(1)) if mm else None
    rows.append((m.group(1), d))
dem = subprocess.run(["c++filt"], input="\n".join(r[0] for r in rows), capture_output=True, text=True).stdout.split("\n")
for (n, d), dn in zip(rows, dem):
    if all(f in dn for f in filters):
        print(dn[:230])
        print("    ", d)
