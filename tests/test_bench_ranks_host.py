"""CPU tier: bench.py's rank body with TWO ranks (the code path the 8-GPU scaling run takes: one process per rank, barrier +
max-over-ranks timing, archives gathered on rank 0 / one mesh's streams spread over the ranks), on gloo with the oracle as the coder
(tests/fake_hip_api.py).  What is checked: both ranks finish, rank 0 prints exactly one JSON line for n_gpus = 2, the archive it
reports is the reference's golden archive of the mesh (sha256, tests/golden/hashes.json) in both sharding modes."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(shard, mode="roundtrip", ranks=2, W=1000, H=1000):
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(ranks), "--steps", "1", "--warmup", "1",
           "--W", str(W), "--H", str(H), "--shard", shard, "--mode", mode, "--backend", "gloo", "--api", "tests.fake_hip_api", "--no-cpu-baseline"]
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""), OMP_NUM_THREADS="1")
    p = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout.decode()[-2000:]
    return json.loads(lines[0])


@pytest.mark.parametrize("shard", ["meshes", "streams"])
def test_rank_body_world2(shard, native_libs):
    j = _run(shard)
    assert j["n_gpus"] == 2 and j["steps"] == 1 and j["unit"] == "GB/s" and j["value"] > 0
    assert "HOST REHEARSAL" in j["data"]
    assert j["scaling"] == ("weak" if shard == "meshes" else "strong")
    # rank 0's archive is the golden one of grid(1000, 1000): written whole by rank 0 ('meshes'), or assembled on rank 0 from
    # payloads both ranks encoded ('streams')
    assert j["config"]["parity"] == "sha256 == reference golden", j["config"]
    golden = json.load(open(os.path.join(ROOT, "tests", "golden", "hashes.json")))["grid_1000x1000"]
    assert j["config"]["archive_bytes_rank0"] == golden["bytes"] if "bytes" in golden else True
    # the self-diagnosis of an N > 1 line: what the backend says about the job, one row per rank
    rk = j["ranks"]
    assert rk["collective"]["backend"] == "gloo" and rk["collective"]["world_size_as_reported"] == 2
    assert [r["rank"] for r in rk["per_rank"]] == [0, 1] and len({r["pid"] for r in rk["per_rank"]}) == 2
    for r in rk["per_rank"]:
        assert r["roundtrip_ok"] is True and r["rank_ms_per_step"] > 0
        assert {"mesh_seed", "archive_bytes", "archive_sha256", "parity", "gathered_bytes_seen", "rank_encode_ms", "rank_gather_ms", "rank_decode_ms"} <= set(r)
    if shard == "meshes":
        assert rk["per_rank"][0]["archive_sha256"] == golden["sha256"][:16] and rk["per_rank"][0]["mesh_seed"] != rk["per_rank"][1]["mesh_seed"]
        assert rk["per_rank"][0]["gathered_bytes_seen"] >= rk["per_rank"][0]["archive_bytes"] + rk["per_rank"][1]["archive_bytes"]
    # the keys of the driver's contract close the line (its stdout_tail keeps the end)
    keys = list(j)
    assert keys.index("value") > keys.index("kernels") and keys[-1] in ("roofline", "cpu_baseline")


@pytest.mark.parametrize("ranks", [2, 3])
def test_decode_mixed_rank_body(ranks, native_libs):
    """BASELINE configs[4]'s N-GPU form (bench.py --mode decode-mixed): every rank builds and decodes an archive of its own kind (rank 0
    grid, rank 1 walk, rank 2 multi: float / double components, u32 / u64 planes), no collective; rank 0 prints one line whose decoded
    bytes are the sum over the ranks and whose own archive is the reference's golden."""
    W, H = 1000, 1000
    j = _run("meshes", mode="decode-mixed", ranks=ranks, W=W, H=H)
    assert j["n_gpus"] == ranks and j["unit"] == "GB/s" and j["value"] > 0 and j["scaling"] == "weak"
    assert j["config"]["mode"] == "decode-mixed" and "HOST REHEARSAL" in j["data"]
    n = W * H
    raw = [n * 12 + 2 * n * 12, n * 12 + 2 * n * 12, 2 * n * 24 + n * 8 + 2 * n * 24]
    assert j["config"]["decoded_bytes_all_ranks"] == sum(raw[:ranks])
    assert j["config"]["parity_rank0"] == "sha256 == reference golden"
    rk = j["ranks"]
    assert rk["collective"]["world_size_as_reported"] == ranks and [r["rank"] for r in rk["per_rank"]] == list(range(ranks))
    assert [r["kind"] for r in rk["per_rank"]] == ["grid", "walk", "multi"][:ranks]
    assert all(r["parity"] == "sha256 == reference golden" for r in rk["per_rank"])


def test_committed_pmc_summary_parses():
    """roofline.traffic comes from a tools/pmc_summary.py text (live, or the newest committed one): the parser must read the
    summaries under profiles/, template kernel names included"""
    import glob
    sys.path.insert(0, ROOT)
    import bench
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_fpc32_encode_hbm_traffic_pmc.txt")))
    assert files
    t = bench._pmc_file_traffic(files[-1])
    assert t and 1.2e9 < t < 3e9, t
    # the double encoder's summary (config3.roofline.traffic): every kernel of the stream, per k64_sizes dispatch
    files = [f for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_fpc64_encode_hbm_traffic_pmc.txt"))) if "_before_" not in f]
    assert files
    t = bench._pmc_file_traffic(files[-1], family=None, per_launch_of=("k64_sizes",))
    assert t and 5e9 < t < 60e9, t
