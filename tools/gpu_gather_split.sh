# the gather per component (one launch each): bash tools/gpu_gather_split.sh
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/gsplit
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for m in grid walk; do
  PERF_PLACE=0 PERF_GATHER_ALL=0 timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/enc -- python $R/tools/perf_fpc32.py $m > $O/$m.log 2>&1
  echo "## $m"
  python - <<PY
import csv, glob
rows=[]
for f in glob.glob("$O/enc/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_fpc32_gather" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3))
rows.sort()
d=[x[1] for x in rows]
print("gather launches (us), three per encode (x, y, z):", " ".join("%.1f" % v for v in d[-9:]))
PY
  rm -rf $O/enc
done
