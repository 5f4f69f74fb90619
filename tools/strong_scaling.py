"""Strong-scaling prediction of BASELINE configs[2] on one device: bench.also_strong_scaling as a tool.  Dev tool (GPU box).
   usage: python tools/strong_scaling.py [reps]"""
import json, os, sys, types
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import numpy as np, torch
import bench
args = types.SimpleNamespace(window=10, extended=int(os.environ.get('EXT', '1')))
print(json.dumps(bench.also_strong_scaling(args, torch, np, reps=int(sys.argv[1]) if len(sys.argv) > 1 else 7), indent=1))
