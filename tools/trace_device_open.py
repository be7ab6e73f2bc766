"""Opens tests/golden/allstreams.trc (18 streams, 44 component frames) from DEVICE memory and walks all of its framing
(type, count and size fields of every stream) without decoding anything.  Run under `rocprofv3 --hip-trace --stats` once with
`upload` (only the upload of the archive) and once with `open`: the difference in hipMemcpy* calls is what the readers need
to see the framing (round 2: one copy per field, ~100 for this file)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from trico_amd import api

mode = sys.argv[1] if len(sys.argv) > 1 else "open"
blob = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "allstreams.trc"), "rb").read()
L = api.lib()
assert L.trico_hip_available() == 1
t = torch.frombuffer(bytearray(blob), dtype=torch.uint8).cuda()
torch.cuda.synchronize()
if mode == "open":
    r = api.Archive.open_for_reading(t)
    infos = api.list_streams(r)
    n = 0
    while r.get_next_stream_type() != 0:
        assert r.skip_next_stream() == 1
        n += 1
    r.close()
    print("streams listed", len(infos), "skipped", n)
