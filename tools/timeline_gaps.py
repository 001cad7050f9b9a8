import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
names=[r["Kernel_Name"] for r in rows]
st=[i for i,n in enumerate(names) if n.startswith("zh_stitch")]
if not st: sys.exit("no zh_stitch launch in the trace")
a=st[-2]+1 if len(st)>1 else 0; b=st[-1]+1
last=rows[a:b]
t0=int(last[0]["Start_Timestamp"])
prev_end=t0
for r in last:
    s=int(r["Start_Timestamp"]);e=int(r["End_Timestamp"])
    n=r["Kernel_Name"].replace("void ","").split("(")[0][:26]
    print("%8.3f dur %7.3f gap %7.3f %s"%((s-t0)/1e6,(e-s)/1e6,(s-prev_end)/1e6,n))
    prev_end=max(prev_end,e)
