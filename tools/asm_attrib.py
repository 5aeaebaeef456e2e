#!/usr/bin/env python3
"""development helper: attributes the vector-ALU instructions of one kernel in a `hipcc -S -g` listing to source
functions (by .loc file:line -> enclosing function found with a crude scan of the source).
usage: asm_attrib.py listing.s <kernel symbol substring> [valu|salu|all] [first last]   (listing lines as asm_loops.py prints them)"""
import collections
import os
import re
import sys

txt = open(sys.argv[1]).read().split("\n")
kind = sys.argv[3] if len(sys.argv) > 3 else "valu"
pat = {"valu": r"\s+v_", "salu": r"\s+s_", "all": r"\s+[vs]_|\s+(global|buffer|scratch|flat|ds)_"}[kind]
files = {}
for l in txt:
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"\s+"([^"]*)"', l)
    if m:
        files[int(m.group(1))] = os.path.join(m.group(2), m.group(3))
start = next(i for i, l in enumerate(txt) if l.startswith("_ZN") and sys.argv[2] in l and ":" in l)
end = next(i for i in range(start + 1, len(txt)) if ".Lfunc_end" in txt[i])
cur = None
hist = collections.Counter()
lo, hi = (int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else (start, end)
for n, l in enumerate(txt[start:end], start):
    m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", l)
    if m:
        cur = (int(m.group(1)), int(m.group(2)))
        continue
    if re.match(pat, l) and cur and lo <= n <= hi:
        hist[cur] += 1
# enclosing function per source line
func_of = {}
for fid, path in files.items():
    if not os.path.exists(path) or "/opt/rocm" in path or "/usr/" in path:
        continue
    name, depth = "?", 0
    for n, l in enumerate(open(path, errors="replace").read().split("\n"), 1):
        if depth == 0 and l and not l[0].isspace() and not l.startswith(("#", "//", "}", "struct", "typedef", "namespace", "enum", "/*", "*")):
            m = re.search(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\(", l.replace("__launch_bounds__(", "__lb__ ").replace("__attribute__(", "__at__ "))
            if m:
                name = m.group(1)
        func_of[(fid, n)] = name
        if l.startswith("namespace") or "// namespace" in l or l.startswith('extern "C"'):
            continue
        depth += l.count("{") - l.count("}")
        if depth == 0 and l.startswith("}"):
            name = "?"
byfunc = collections.Counter()
for (fid, line), c in hist.items():
    byfunc[(os.path.basename(files.get(fid, "?")), func_of.get((fid, line), "?"))] += c
tot = sum(byfunc.values())
print("total", kind, tot)
for (f, fn), c in byfunc.most_common(60):
    print("%6d %5.1f%%  %s:%s" % (c, 100.0 * c / tot, f, fn))
