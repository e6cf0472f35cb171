#!/usr/bin/env python3
"""Summarise `make -C denovo_kmer_amd/csrc resources` (hipcc -Rpass-analysis=kernel-resource-usage): one line per kernel
with SGPRs, VGPRs, scratch, occupancy and LDS.  usage: tools/kernel_resources.py [substring ...]"""
import re
import subprocess
import sys

out = subprocess.run(["make", "-C", "denovo_kmer_amd/csrc", "resources"], capture_output=True, text=True).stderr
name, d = None, {}
for l in out.splitlines():
    m = re.search(r"Function Name: (\S+)", l)
    if m:
        name = m.group(1)
        d[name] = {}
    for k, short in (("TotalSGPRs", "sgpr"), ("VGPRs", "vgpr"), (r"ScratchSize \[bytes/lane\]", "scratch"),
                     (r"Occupancy \[waves/SIMD\]", "occ"), (r"LDS Size \[bytes/block\]", "lds")):
        m = re.search(r" " + k + r": (\d+)", l)
        if m and name:
            d[name][short] = int(m.group(1))
names = list(d)
dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.splitlines()
for n, dm in zip(names, dem):
    dm = re.sub(r"\(.*", "", dm).replace("void dk::", "")
    if len(sys.argv) == 1 or any(x in dm for x in sys.argv[1:]):
        print("%-90s %s" % (dm[:90], d[n]))
