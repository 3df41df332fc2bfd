"""The FID feature pass alone (bench.fid50k_record) for a profiler:   python tools/fid_prof.py [n_samples] [batch]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench      # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
b = int(sys.argv[2]) if len(sys.argv) > 2 else 250
torch.cuda.set_device(0)
rec = bench.fid50k_record(torch.device("cuda", 0), n, b)
print(json.dumps({k: rec[k] for k in ("seconds", "value", "roofline")}))
