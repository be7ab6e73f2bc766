"""config 3 (double vertices + double normals + float uv) alone: encode / decode seconds, for experiments on the double decoder"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from trico_amd import api, meshgen
W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (10000, 5000)
dev = torch.device("cuda:0")
r = bench.config3_block(api, meshgen, dev, W, H)
print(json.dumps({k: r[k] for k in ("encode_s", "decode_s", "decode_GBps", "parity")}))
