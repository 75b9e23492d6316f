import csv, collections, sys
rows=[r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
idx=[i for i,r in enumerate(rows) if "k_loop_init" in r["Kernel_Name"]]
rows=rows[idx[-1]-6:]
d=collections.defaultdict(lambda:[0,0])
for r in rows:
    k=r["Kernel_Name"].split("(")[0][:50]
    d[k][0]+=1; d[k][1]+=int(r["End_Timestamp"])-int(r["Start_Timestamp"])
for k,v in sorted(d.items(), key=lambda kv:-kv[1][1]): print(k.ljust(52), v[0], round(v[1]/1e6,3),"ms")
for name in ("k_loop_eval","k_loop_first","k_loop_accept","k_loop_apply"):
    ev=[r for r in rows if name in r["Kernel_Name"]]
    print(name, [ (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))//1000 for r in ev][:80])
