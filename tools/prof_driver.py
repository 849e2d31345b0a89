"""Small driver for rocprofv3 passes: nothing but the compression launches after setup.
    python3 tools/prof_driver.py <config> <tie_mode> <prompts> <mode: calls|batch> [layers]
    rocprofv3 --pmc FETCH_SIZE --kernel-trace -- python3 tools/prof_driver.py c2 torch_cpu 6 batch   (one --pmc group per run)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
dev = torch.device("cuda:0")
name = sys.argv[1] if len(sys.argv) > 1 else "c2"
cfg = bench.CONFIGS[name]
tie = sys.argv[2] if len(sys.argv) > 2 else "torch_cpu"
n = int(sys.argv[3]) if len(sys.argv) > 3 else 8
mode = sys.argv[4] if len(sys.argv) > 4 else "batch"
bench.LAYERS = int(sys.argv[5]) if len(sys.argv) > 5 else cfg.get("layers", 32)
if cfg["method"] == "h2o":
    mode = "calls"
ks = bench.layer_budgets(cfg)
p = bench.Prompt(cfg, dev, tie, mode, bench.make_inputs(cfg, dev, 0), ks, 1)
for _ in range(n):
    p.run()
torch.cuda.synchronize()
print("done", name, mode, bench.LAYERS, "layers x", n)
