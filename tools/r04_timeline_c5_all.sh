O=gpurun_out/r04e; mkdir -p $O; cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --memory-copy-trace -d $O/kt_c5 -o kt --output-format csv -- python3 bench.py --config 5 --files 262144 --profile-run --steps 2 --warmup 1 > $O/kt_c5.json 2> $O/kt_c5.err
python tools/timeline.py $(find $O/kt_c5 -name "*kernel_trace.csv" | head -1) $O/timeline_c5_all.txt all
ls $O/kt_c5/*; head -3 $(find $O/kt_c5 -name "*memory_copy_trace.csv" | head -1)
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/r04e/kt_c5/**/*memory_copy_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
k=glob.glob('gpurun_out/r04e/kt_c5/**/*kernel_trace.csv',recursive=True)[0]
kr=list(csv.DictReader(open(k))); kr.sort(key=lambda r:int(r["Start_Timestamp"]))
st=[r for r in kr if r["Kernel_Name"].startswith("zh_stitch")]
a=int(st[-2]["End_Timestamp"]); b=int(st[-1]["End_Timestamp"])
for r in rows:
    s=int(r["Start_Timestamp"]); e=int(r["End_Timestamp"])
    if s>=a and s<=b+3000000: print("%9.3f %9.3f %8.3f %s %s"%((s-a)/1e6,(e-a)/1e6,(e-s)/1e6,r.get("Direction"),r.get("Bytes", r.get("Size"))))
PY
rm -rf $O/kt_c5
cat $O/timeline_c5_all.txt
