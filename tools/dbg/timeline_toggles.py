import csv,sys
rows=[r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
idx=[i for i,r in enumerate(rows) if "list_toggle" in r["Kernel_Name"]]
for i0 in idx[-3:-1]:
    for r in rows[i0:i0+4]:
        s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
        n=r["Kernel_Name"]
        n=[k for k in ("walk_toggles","find_toggle","list_toggle","local_toggles","copyBuffer","fillBuffer") if k in n]
        print("%-14s dur %7.1f us" % (n[0] if n else "?", (e-s)/1e3))
