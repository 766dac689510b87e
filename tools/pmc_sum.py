import csv, collections, sys
names=["edge_bwd","edge_fwd","pw_bwd_main","pw_w1_nodesums","pw_w1_classrows","pw_fwd","gather_sparse","gather_sums","winners_mark","blk_bwd_pre","blk_bwd_post","node_fwd","reduce_partials","graph_sweep","head_bwd","match_greedy"]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter(); dur=collections.defaultdict(float); seen=set()
for row in csv.DictReader(open(sys.argv[1])):
    k = next((x for x in names if x in row["Kernel_Name"]), None)
    if not k: continue
    agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
    if row["Dispatch_Id"] not in seen:
        seen.add(row["Dispatch_Id"]); n[k]+=1; dur[k]+= (int(row["End_Timestamp"])-int(row["Start_Timestamp"]))/1e3
for k, v in sorted(agg.items(), key=lambda kv:-dur[kv[0]]):
    print("%-16s n=%3d avg_us=%8.1f " % (k, n[k], dur[k]/n[k]) + " ".join("%s=%.4g" % (c.replace("SQ_","").replace("_sum",""), x/n[k]) for c,x in sorted(v.items())))
