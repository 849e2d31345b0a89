"""TEST INFRASTRUCTURE.  Builds oracle/gen_killers.cpp, runs it and writes tests/golden/killer_<case>.npz:
  codes  int32 [n]   adversarial score codes (larger = higher score; see killer_adversary.h)
  k      int         the topk budget the row was built against
  want   int64 [k]   torch.topk(codes.float(), k).indices on the CPU of the generating machine (torch-CPU topk = the
                     reference's selection, pyramidkv_utils.py:334), which the oracle's libstdc++ calls must reproduce.
Usage: python oracle/gen_killers.py"""
import os, subprocess, tempfile
import numpy as np, torch

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")

with tempfile.TemporaryDirectory() as td:
    exe = os.path.join(td, "gen_killers")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(HERE, "gen_killers.cpp")])
    text = subprocess.run([exe], check=True, capture_output=True, text=True)
print(text.stderr, end="")
for line in text.stdout.splitlines():
    f = line.split()
    name, n, k = f[0], int(f[1]), int(f[2])
    codes = np.array(f[3:], dtype=np.int32)
    assert codes.shape == (n,)
    want = torch.topk(torch.from_numpy(codes).float(), k, dim=-1).indices.numpy()
    np.savez_compressed(os.path.join(OUT, f"killer_{name}.npz"), codes=codes, k=np.int64(k), want=want)
    print(f"killer_{name}.npz: n={n} k={k} distinct={len(np.unique(codes))}")
