"""Instruction histogram per kernel of a hipcc --cuda-device-only -S listing:  python tools/isa_hist.py file.s [name-filter]"""
import collections
import re
import sys

txt = open(sys.argv[1]).read().split("\n")
flt = sys.argv[2] if len(sys.argv) > 2 else ""
name, c = None, None
out = []
for line in txt:
    m = re.match(r"^(_Z\w+):", line)
    if m:
        name, c = m.group(1), collections.Counter()
        out.append((name, c))
        continue
    if name is None:
        continue
    if line.startswith(".Lfunc_end"):
        name = None
        continue
    m = re.match(r"^\s+((?:v|s|ds|global|buffer|scratch|flat)_[a-z0-9_]+)", line)
    if m:
        c[m.group(1)] += 1
for name, c in out:
    if flt in name:
        print(name[:90], sum(c.values()))
        print("    " + ", ".join("%s %d" % kv for kv in c.most_common(28)))
