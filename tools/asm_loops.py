#!/usr/bin/env python3
"""development helper: lists the loops of one kernel in a hipcc -S listing with their instruction mix
usage: asm_loops.py listing.s <substring of the kernel symbol>"""
import re
import sys
txt = open(sys.argv[1]).read().split("\n")
start = next(i for i, l in enumerate(txt) if l.startswith("_ZN") and sys.argv[2] in l and l.rstrip().endswith(":") is False and ":" in l)
end = next(i for i in range(start + 1, len(txt)) if txt[i].startswith("\t.section") or txt[i].startswith("\t.end_amdhsa_kernel") or ".Lfunc_end" in txt[i])
lines = txt[start:end]
labels = {}
for i, l in enumerate(lines):
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        labels[m.group(1)] = i
def mix(body):
    c = lambda pat: sum(1 for x in body if re.match(pat, x))
    return "valu %d salu %d vmem %d lds %d" % (c(r"\s+v_"), c(r"\s+s_"), c(r"\s+(global|buffer|scratch|flat)_"), c(r"\s+ds_"))
print("kernel lines", start, end, "total:", mix(lines))
seen = set()
for i, l in enumerate(lines):
    m = re.search(r"s_c?branch\S*\s+(\.LBB\d+_\d+)", l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        a = labels[m.group(1)]
        print(m.group(1), "lines %d-%d" % (start + a + 1, start + i + 1), "len", i - a, mix(lines[a:i + 1]))
