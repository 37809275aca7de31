"""Skeleton of a kernel's ISA: the order of memory requests (L), vmcnt waits (wN), matrix instructions (M, runs collapsed), barriers (|)
and branches (^) - enough to see whether a batch of requests is in flight together or the scheduler moved each next to its use
("w0 MMLML w0 MMLML ...": one exposed round trip per k step).  Also lists kernels with lone requests issued right after a full drain.
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -S --cuda-device-only csrc/<file>.hip -o /tmp/<file>.s
  python tools/isa_skeleton.py /tmp/<file>.s [substring of the mangled kernel name]"""
import re
import sys

s = open(sys.argv[1]).read()
want = sys.argv[2] if len(sys.argv) > 2 else None
seen = set()
for nm in re.findall(r"^(_Z\w+):\s*(?:;.*)?$", s, re.M):
    if nm in seen:
        continue
    seen.add(nm)
    i = re.search(r"^" + re.escape(nm) + r":", s, re.M).end()
    j = s.find("s_endpgm", i)
    if j < 0 or re.search(r"^_Z\w+:", s[i:j], re.M):
        continue
    seq = []
    for line in s[i:j].split("\n"):
        t = line.strip()
        if t.startswith(("global_load", "buffer_load")):
            seq.append("L")
        elif t.startswith("s_waitcnt") and "vmcnt" in t:
            seq.append("w" + re.search(r"vmcnt\((\d+)\)", t).group(1) + " ")
        elif t.startswith("v_mfma"):
            seq.append("M")
        elif t.startswith("s_barrier"):
            seq.append("|")
        elif t.startswith(("s_cbranch", "s_branch")):
            seq.append("^")
    st = "".join(seq)
    lone = len(re.findall(r"w0 \^?L(?=w0 |M)", st))
    if want is None:
        if st.count("L") > 4:
            print("%-90s requests %4d  lone after a drain %3d  matrix %d" % (nm[:90], st.count("L"), lone, st.count("M")))
    elif want in nm:
        print(nm)
        print(re.sub(r"M{6,}", lambda m: "M%d" % len(m.group(0)), st))
