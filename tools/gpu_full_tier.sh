# the whole GPU tier, then the bench line
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/full_tier
mkdir -p $O
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1 || { tail -60 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
timeout -k 10 900 python bench.py --steps 3 --warmup 1 > $O/bench.json 2> $O/bench.err || { tail -30 $O/bench.err; exit 1; }
python - <<'PY'
import json,os
j=json.loads(open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/full_tier/bench.json').read().strip().splitlines()[-1])
print("value", j["value"], j["unit"], "ms", j["ms_per_step"], "roofline", j["roofline"]["frac"], j["roofline"]["traffic"])
print("kernels", json.dumps(j["kernels"]))
print("concurrent", json.dumps(j.get("decode_concurrent", {}).get("vs_cpu_all_cores")))
print("rows", [(r["archives"], r["decode_GBps"]) for r in j.get("decode_concurrent", {}).get("results", [])])
print("pcie", json.dumps(j["pcie_inclusive"]))
print("config3", j["config3"]["decode_s"], j["config3"]["encode_s"], "mixed", json.dumps(j.get("config5_mixed"))[:300])
print("other", json.dumps(j.get("other_mesh"))[:600])
PY
