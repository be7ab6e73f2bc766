"""GPU tier: the chunk-speculative LZ4 compressor (k_lz4_chunked.hip) with many small chunks, against the
oracle.  Run in a subprocess because the chunk geometry is read from the environment once per process."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("chunk,warm", [(131072, 70000), (262144, 131072)])
def test_chunked_lz4_matches_oracle(native_libs, chunk, warm):
    env = dict(os.environ)
    env.update({"TRICO_LZ4_CHUNK": str(chunk), "TRICO_LZ4_WARM": str(warm), "TRICO_LZ4_CHUNKED_MIN": "65547"})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_lz4_chunked_worker.py")], env=env,
                       capture_output=True, text=True, timeout=600)
    print(p.stdout)
    print(p.stderr[-2000:])
    assert p.returncode == 0, p.stdout + p.stderr[-2000:]
    assert "exact=False" not in p.stdout and "roundtrip=False" not in p.stdout
