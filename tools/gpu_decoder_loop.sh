# float chain decoder: parity, then the bench line
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/dec_loop
mkdir -p $O
cd $R
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_lowlevel.py tests/test_gpu_batch.py tests/test_gpu_selfcheck.py -m gpu -x -q > $O/pytest.log 2>&1 || { tail -60 $O/pytest.log; exit 1; }
tail -2 $O/pytest.log
timeout -k 10 600 python bench.py --steps 2 --warmup 1 > $O/bench.json 2> $O/bench.err || { tail -30 $O/bench.err; exit 1; }
python - <<'PY'
import json,os
j=json.loads(open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/dec_loop/bench.json').read().strip().splitlines()[-1])
print("value", j["value"], j["unit"], "ms", j["ms_per_step"])
for k in ("decode_model","decode","kernels"):
    if k in j: print(k, json.dumps(j[k])[:600])
PY
