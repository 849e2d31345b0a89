"""Small driver for rocprofv3 counter passes: a few C2 layer calls, single stream, no synthetic-data kernels after setup.
    rocprofv3 --kernel-trace --stats -- python3 tools/prof_driver.py
    rocprofv3 --pmc FETCH_SIZE --kernel-trace -- python3 tools/prof_driver.py        (one --pmc group per run)"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from kvcache_factory_amd import _kvc
dev = torch.device("cuda:0")
cfg = bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "c2"]
bench.LAYERS = 4
plans, ks = bench.build_plans(cfg, dev, sys.argv[2] if len(sys.argv) > 2 else "canonical", expanded=False, n_streams=1)
for _ in range(int(sys.argv[3]) if len(sys.argv) > 3 else 8):
    bench.run_step(plans)
torch.cuda.synchronize()
print("done", len(plans), "layers")
