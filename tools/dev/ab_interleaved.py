"""A/B of development builds that survives a box whose speed drifts: python tools/dev/ab_interleaved.py D rounds lib_a.so lib_b.so ...
Each library runs in a process of its own per round (HC_PROF_MEMBERS members, 2 days), the rounds interleave the libraries
(a b c a b c ...), and the table shows every round -- a drift of the box shows up as a drift of EVERY column."""
import os, re, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
D, rounds, libs = sys.argv[1], int(sys.argv[2]), sys.argv[3:]
res = {l: [] for l in libs}
sha = {}
for r in range(rounds):
    for l in libs:
        p = subprocess.run([sys.executable, os.path.join(R, "tools", "prof_depth.py"), l, D], capture_output=True, text=True, timeout=300)
        m = re.search(r": (\d+) column-days/s .* sha (\w+)", p.stdout)
        if not m:
            print(l, "FAILED", p.stdout[-300:], p.stderr[-300:]); res[l].append(0); continue
        res[l].append(int(m.group(1))); sha.setdefault(l, set()).add(m.group(2))
base = libs[0]
for l in libs:
    v = res[l]
    ratio = [b / a if a else 0 for a, b in zip(res[base], v)]
    print(f"D={D} {os.path.basename(l):22s} " + " ".join(f"{x:7d}" for x in v) + f"   vs {os.path.basename(base)}: " +
          " ".join(f"{x:.3f}" for x in ratio) + f"   sha {sorted(sha.get(l, []))}", flush=True)
