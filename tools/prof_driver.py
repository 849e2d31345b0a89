"""Small driver for rocprofv3 counter passes (no synthetic-data kernels after setup).
    python3 tools/prof_driver.py <config> <tie_mode> <steps> <mode: calls|batch> [layers]
    rocprofv3 --pmc FETCH_SIZE --kernel-trace -- python3 tools/prof_driver.py c2 canonical 6 batch   (one --pmc group per run)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device("cuda:0")
cfg = bench.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "c2"]
tie = sys.argv[2] if len(sys.argv) > 2 else "canonical"
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 8
mode = sys.argv[4] if len(sys.argv) > 4 else "calls"
bench.LAYERS = int(sys.argv[5]) if len(sys.argv) > 5 else (32 if mode == "batch" else 4)
plans, ks = bench.build_plans(cfg, dev, tie, expanded=False, n_streams=1)
if mode == "batch":
    plans = [bench.BatchStep(cfg, dev, tie, plans, ks)]
for _ in range(steps):
    bench.run_step(plans)
torch.cuda.synchronize()
print("done", mode, bench.LAYERS, "layers")
