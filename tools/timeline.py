import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# take the last batch: find last zh_stitch and go back to the previous one
names=[r["Kernel_Name"] for r in rows]
st=[i for i,n in enumerate(names) if n.replace("void ","").split("(")[0]=="zh_stitch"]   # (not zh_stitch_scan)
if not st:   # a command that never stitches (tools/step_dev.py): a batch ends with zh_compact_results
    st=[i for i,n in enumerate(names) if n.replace("void ","").split("(")[0]=="zh_compact_results"]
a=st[-2]+1 if len(st)>1 else 0; b=st[-1]+1
last=rows[a:b]
t0=int(last[0]["Start_Timestamp"])
out=open(sys.argv[2],"w")
for r in last:
    s=int(r["Start_Timestamp"]);e=int(r["End_Timestamp"])
    n=r["Kernel_Name"].replace("void ","").split("(")[0][:22]
    if n.startswith("__amd") and len(sys.argv) < 4: continue   # a third argument: the runtime's copy / fill kernels too
    out.write("%9.3f %9.3f %8.3f %s q=%s\n"%((s-t0)/1e6,(e-t0)/1e6,(e-s)/1e6,n,r.get("Queue_Id","?")))
out.close()
