"""Where does a step's wall time go?  From a rocprofv3 kernel_trace.csv: for the LAST full step (between two consecutive
pack_transpose launches), per kernel name: launches, summed duration; and the time no kernel was running on any queue
(gaps), the union busy time, the critical (main) queue's busy time.  usage: trace_gaps.py <kernel_trace.csv>"""
import collections
import csv
import re
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")))
rows.sort()
packs = [i for i, r in enumerate(rows) if "pack_transpose" in r[2]]
if len(packs) < 3:
    sys.exit("need >= 3 steps in the trace")
a, b = packs[-2], packs[-1]
# a step's side-stream kernels of step i+1 (graph count) interleave: take everything that STARTS inside [t0, t1)
t0, t1 = rows[a][0], rows[b][0]
step = [r for r in rows if t0 <= r[0] < t1]
short = lambda n: re.sub(r"\(anonymous namespace\)::|void |\(.*", "", n)[:40]
agg = collections.OrderedDict()
for s, e, n, q in step:
    k = short(n)
    c = agg.setdefault(k, [0, 0.0, set()])
    c[0] += 1; c[1] += (e - s) / 1e3; c[2].add(q)
print("step wall %.1f us, %d launches" % ((t1 - t0) / 1e3, len(step)))
# union of busy intervals
ivs = sorted((s, e) for s, e, _, _ in step)
busy, cur_s, cur_e = 0, ivs[0][0], ivs[0][1]
for s, e in ivs[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print("some kernel running: %.1f us; nothing running: %.1f us" % (busy / 1e3, (t1 - t0 - busy) / 1e3))
qs = collections.Counter(q for _, _, _, q in step)
for q, n in qs.most_common():
    d = sum(e - s for s, e, _, qq in step if qq == q) / 1e3
    print("queue %s: %d launches, %.1f us of kernels" % (q, n, d))
print("%-42s %5s %10s %9s  queues" % ("kernel", "n", "total us", "avg us"))
for k, (n, d, q) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-42s %5d %10.1f %9.1f  %s" % (k, n, d, d / n, ",".join(sorted(q))))
# gaps on the main queue between consecutive kernels, by predecessor kernel
mainq = qs.most_common(1)[0][0]
mq = [r for r in step if r[3] == mainq]
gaps = collections.defaultdict(lambda: [0, 0.0])
for (s0, e0, n0, _), (s1, e1, n1, _) in zip(mq, mq[1:]):
    g = gaps[short(n0) + " -> " + short(n1)]
    g[0] += 1; g[1] += (s1 - e0) / 1e3
print("gaps on the main queue (end of a kernel -> start of the next), total %.1f us" % sum(v[1] for v in gaps.values()))
for k, (n, d) in sorted(gaps.items(), key=lambda kv: -kv[1][1])[:14]:
    print("  %-70s %4d %8.1f us  (%.1f avg)" % (k, n, d, d / n))
