"""The .lrn loader / writer leg of bench.py alone.  usage: python3 experiments/lrn_prof.py"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch; torch.cuda.init(); torch.zeros(1, device="cuda")
import bench
from abawaca_amd import _lib
ctx = _lib.Context(0)
print(json.dumps(bench.lrn_bench(ctx, False), indent=1))
