"""Per-kernel, per-launch averages of a rocprofv3 --pmc counter_collection.csv (one pass).  usage: pmc_sum.py <csv> [--json]"""
import collections
import csv
import json
import sys

# substring of the kernel symbol -> name used in the profile summaries / profiles/*_traffic.json (bench.py kernel classes)
NAMES = [("edge_bwd_w", "edge_bwd"), ("edge_fwd_w", "edge_fwd"), ("pw_bwd_bf", "pw_bwd_main"), ("pw_bwd_main", "pw_bwd_main"), ("pw_w1_nodesums", "pw_w1_nodesums"),
         ("pw_w1_classrows", "pw_w1_classrows"), ("pw_fwd", "pw_fwd"), ("gather_winners", "gather_winners"),
         ("winners_mark", "winners_mark"), ("winners_ties", "winners_ties"), ("winner_positions", "winner_positions"),
         ("list_fill", "list_fill"), ("list_count", "list_count"), ("blk_bwd_node", "node_bwd"), ("node_fwd", "node_fwd"),
         ("reduce_partials", "reduce_partials"), ("graph_sweep", "graph_sweep"), ("head_bwd", "head_bwd"),
         ("match_greedy", "match_greedy"), ("edge_geometry", "edge_geometry"), ("roi_pool_fwd", "roi_pool_fwd"),
         ("roi_pool_bwd_pixel", "roi_pool_bwd_pixel"), ("roi_pool_bwd_block", "roi_pool_bwd_block"), ("roi_pool_bwd_atomic", "roi_pool_bwd_atomic"), ("imfeat", "imfeat")]


def summarise(path):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    n = collections.Counter(); dur = collections.defaultdict(float); seen = set()
    for row in csv.DictReader(open(path)):
        k = next((nm for sub, nm in NAMES if sub in row["Kernel_Name"]), None)
        if not k:
            continue
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
        if row["Dispatch_Id"] not in seen:
            seen.add(row["Dispatch_Id"]); n[k] += 1
            dur[k] += (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3
    return {k: dict(launches=n[k], avg_us=dur[k] / n[k], **{c.replace("SQ_", "").replace("_sum", ""): x / n[k] for c, x in v.items()})
            for k, v in agg.items()}


if __name__ == "__main__":
    res = summarise(sys.argv[1])
    if "--json" in sys.argv:
        print(json.dumps(res))
    else:
        for k, v in sorted(res.items(), key=lambda kv: -kv[1]["avg_us"] * kv[1]["launches"]):
            print("%-16s n=%3d avg_us=%8.1f " % (k, v["launches"], v["avg_us"]) +
                  " ".join("%s=%.4g" % (c, x) for c, x in sorted(v.items()) if c not in ("launches", "avg_us")))
