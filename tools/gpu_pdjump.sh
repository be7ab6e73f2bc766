# durations of the 20 jump rounds of one LZ4 decode (and of the other k_pd_* kernels): bash tools/gpu_pdjump.sh [grid|walk]...
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pdjump
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for m in "$@"; do
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $O/tr -- python $R/tools/perf_lz4.py $m > $O/$m.log 2>&1
  echo "## $m: $(grep 'decode iter 1' $O/$m.log)"
  python - <<PY
import csv, glob
rows=[]
for f in glob.glob("$O/tr/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_pd_" in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), r["Kernel_Name"].split("(")[0].split("::")[-1], (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3))
rows.sort()
last=[x for x in rows][-27:]
print(" ".join("%s:%.0f" % (n.replace("k_pd_",""), d) for _, n, d in last))
PY
  rm -rf $O/tr
done
