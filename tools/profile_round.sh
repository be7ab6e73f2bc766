# Collects the profiles committed under profiles/ (layout of rounds 3 and 4).  usage (GPU box): bash tools/profile_round.sh <tag>
#   <tag>_bench_config2.json / _kernel_stats.txt      bench.py --quick under rocprofv3 --kernel-trace --stats
#   <tag>_bench_config2_unprofiled.json               the full bench.py line, no profiler attached
#   <tag>_fpc32_encode_kernel_stats.txt               the float-vertex encoder ALONE (tools/perf_fpc32.py grid): per-kernel durations
#   <tag>_fpc32_encode_hbm_traffic_pmc.txt            --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes, same program
#   <tag>_fpc32_encode_sq_counters.txt                SQ instruction / cycle / LDS counters, same program, both meshes
#   <tag>_fpc64_encode_kernel_stats.txt / _hbm_traffic_pmc.txt   the double encoder alone (tools/perf_fpc64.py): per-kernel durations, FETCH / WRITE
#   <tag>_lz4_kernel_stats.txt                        tools/perf_lz4.py grid and walk: per-kernel durations of the LZ4 codec
#   <tag>_device_archive_open_hip_api.txt             hipMemcpy* calls to walk the framing of a device-resident archive
set -e
TAG=${1:-r06}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_$TAG
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
echo "== full bench, no profiler"
timeout -k 10 600 python $R/bench.py --steps 3 --warmup 1 > $O/${TAG}_bench_config2_unprofiled.json 2> $O/bench_unprofiled.err
echo "== kernel trace of bench.py --quick"
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench -- python $R/bench.py --quick --steps 3 --warmup 1 > $O/${TAG}_bench_config2.json 2> $O/bench.err
python $R/tools/prof_summary.py $O/bench > $O/${TAG}_bench_config2_kernel_stats.txt
echo "== encoder alone: kernel trace, both meshes"
{ for m in grid walk; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/enc_$m -- python $R/tools/perf_fpc32.py $m > $O/enc_$m.log 2>&1
  echo "# rocprofv3 --kernel-trace --stats -- python tools/perf_fpc32.py $m: the float-vertex encoder alone on the 50 M config-2 vertices (6 encodes,"
  echo "# payloads gathered to a device buffer with ONE launch for the three components, as the archive writer does; no decoder, no self-check traffic beside it)"
  grep "kernel span" $O/enc_$m.log
  python $R/tools/prof_summary.py $O/enc_$m
  rm -rf $O/enc_$m
done; } > $O/${TAG}_fpc32_encode_kernel_stats.txt
echo "== PMC: HBM traffic of the float encoder (separate passes, both meshes)"
{ echo "# rocprofv3 --pmc FETCH_SIZE (own pass) and --pmc WRITE_SIZE (own pass), tools/perf_fpc32.py {grid|walk} (50M float xyz vertices), per dispatch, unit KB"
  echo "# gfx950: FETCH_SIZE reports 1/2 of the bytes of streaming reads (MI355X_MICROARCH.md, HBM section) -> doubled in DESIGN.md and bench.py"
  for m in grid walk; do
    echo "## $m"
    for c in FETCH_SIZE WRITE_SIZE; do
      timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_${c}_$m -- python $R/tools/perf_fpc32.py $m > $O/pmc_${c}_$m.log 2>&1
      python $R/tools/pmc_summary.py $O/pmc_${c}_$m
      rm -rf $O/pmc_${c}_$m
    done
  done; } > $O/${TAG}_fpc32_encode_hbm_traffic_pmc.txt
echo "== PMC: SQ counters, both meshes"
{ echo "# rocprofv3 SQ counters of the float encoder's kernels (tools/perf_fpc32.py {grid|walk}), per dispatch, six passes per mesh"
  for m in grid walk; do
    echo "## $m"
    i=0
    for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_BRANCH SQ_WAVES" \
               "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_BUSY_CU_CYCLES" \
               "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_IFETCH SQ_BUSY_CYCLES" \
               "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_DATA_FIFO_FULL"; do
      i=$((i+1))
      timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/pmc_sq$i -- python $R/tools/perf_fpc32.py $m > $O/pmc_sq$i.log 2>&1
      python $R/tools/pmc_summary.py $O/pmc_sq$i | grep -A7 "^k_fpc32"
      rm -rf $O/pmc_sq$i
    done
  done; } > $O/${TAG}_fpc32_encode_sq_counters.txt
echo "== double encoder: kernel trace and HBM traffic"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/enc64 -- python $R/tools/perf_fpc64.py > $O/enc64.log 2>&1
{ echo "# rocprofv3 --kernel-trace --stats -- python tools/perf_fpc64.py: the double encoder alone on the two vec3 double streams of multi(10000,5000)"
  echo "# (50 M double vertices, 50 M double normals; 4 x 2 encodes)"
  grep "kernel span" $O/enc64.log
  python $R/tools/prof_summary.py $O/enc64
  rm -rf $O/enc64; } > $O/${TAG}_fpc64_encode_kernel_stats.txt
{ echo "# rocprofv3 --pmc FETCH_SIZE (own pass) and --pmc WRITE_SIZE (own pass), tools/perf_fpc64.py, per dispatch, unit KB (FETCH_SIZE x 2 on gfx950)"
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc64_$c -- python $R/tools/perf_fpc64.py > $O/pmc64_$c.log 2>&1
    python $R/tools/pmc_summary.py $O/pmc64_$c
    rm -rf $O/pmc64_$c
  done; } > $O/${TAG}_fpc64_encode_hbm_traffic_pmc.txt
echo "== LZ4 codec: kernel traces"
for m in grid walk; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/lz4_$m -- python $R/tools/perf_lz4.py $m > $O/lz4_$m.log 2>&1
done
{ for m in grid walk; do
    echo "# rocprofv3 --kernel-trace --stats -- python tools/perf_lz4.py $m: byte planes + LZ4 of the 100 M config-2 triangles, 3 encodes + 2 decodes"
    grep -E "encode iter 2|decode iter 1" $O/lz4_$m.log
    python $R/tools/prof_summary.py $O/lz4_$m
  done; } > $O/${TAG}_lz4_kernel_stats.txt
echo "== device-resident archive: HIP API calls to read the framing"
for m in upload open; do
  timeout -k 10 200 rocprofv3 --hip-trace --stats --output-format csv -d $O/hip_$m -- python $R/tools/trace_device_open.py $m > $O/hip_$m.log 2>&1
done
{ echo "# rocprofv3 --hip-trace --stats -- python tools/trace_device_open.py {upload|open}: tests/golden/allstreams.trc (18 streams, 44 frames) in device memory."
  echo "# 'upload' = torch moves the archive to the GPU and nothing else; 'open' = the same + trico_open_archive_for_reading on the device pointer,"
  echo "# trico_hip_list_streams and a skip of every stream (all type / count / size fields are read).  Calls per HIP memcpy entry point:"
  for m in upload open; do
    echo "## $m"; cat $O/hip_$m.log | grep -v "^[WE]2" | tail -2
    python - $O/hip_$m <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*hip_api_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "emcpy" in r["Name"] or "LaunchKernel" in r["Name"] or "StreamSynchronize" in r["Name"]:
            print("   %-40s calls %s" % (r["Name"], r["Calls"]))
PY
  done; } > $O/${TAG}_device_archive_open_hip_api.txt
rm -rf $O/bench $O/lz4_grid $O/lz4_walk $O/hip_upload $O/hip_open
ls -la $O
