"""Instruction histogram of a range of lines of an ISA listing (hipcc -S): python tools/isa_hist.py file.s first last
Classes: mfma / valu / salu / lds / vmem / smem / wait / branch / other; prints the most frequent mnemonics of each."""
import collections
import sys


def classify(m):
    if m.startswith("v_mfma"): return "mfma"
    if m.startswith(("ds_",)): return "lds"
    if m.startswith(("global_", "buffer_", "flat_", "scratch_")): return "vmem"
    if m.startswith(("s_load", "s_buffer_load")): return "smem"
    if m.startswith(("s_waitcnt", "s_nop", "s_barrier", "s_sleep", "s_setprio", "s_sched")): return "wait"
    if m.startswith(("s_cbranch", "s_branch")): return "branch"
    if m.startswith("v_"): return "valu"
    if m.startswith("s_"): return "salu"
    return "other"


def main():
    path, a, b = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    by = collections.defaultdict(collections.Counter)
    for i, line in enumerate(open(path), 1):
        if i < a or i > b: continue
        t = line.split(";")[0].strip()
        if not t or t.startswith(".") or t.endswith(":"): continue
        m = t.split()[0]
        by[classify(m)][m] += 1
    for c in ("mfma", "valu", "salu", "lds", "vmem", "smem", "wait", "branch", "other"):
        n = sum(by[c].values())
        if n: print("%-6s %5d   %s" % (c, n, ", ".join("%s x%d" % kv for kv in by[c].most_common(14))))


if __name__ == "__main__":
    main()
