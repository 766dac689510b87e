"""Whole-step A/B of library builds inside ONE process is not possible (one library per process): this runs tools/quick_bench.py's two
shapes under each library in turn, several rounds, and prints the minima.  python tools/ab_step.py <rounds> <name> [<name> ...]  ("main" or tools/ab/<name>.so)"""
import os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rounds = int(sys.argv[1]); names = sys.argv[2:]
best = {}
for r in range(rounds):
    for n in names:
        env = dict(os.environ)
        if n != "main":
            env["GNET_LIB_AB"] = os.path.join(root, "tools", "ab", n + ".so")
        out = subprocess.run([sys.executable, os.path.join(root, "tools", "quick_bench.py"), "30"], env=env, capture_output=True, text=True).stdout
        for line in out.splitlines():
            if line.startswith("images/step"):
                k = (n, line.split()[1].rstrip(":"))
                ms = float(line.split("det/s")[1].split("ms/step")[0])
                best.setdefault(k, []).append(ms)
for k, v in sorted(best.items()):
    print("%-12s images/step %s: %s  min %.4f ms" % (k[0], k[1], " ".join("%.4f" % x for x in v), min(v)))
