import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import qrkit_amd
import bench
ctx = qrkit_amd.Context(0)
dev = torch.device("cuda", 0)
print("alone:", bench.angular_config3(ctx, dev, torch, np)["compute_ms"], flush=True)
torch.cuda.empty_cache()
m = bench.mixed_share(ctx, dev, torch, np)
print("mixed:", m["ms_best"], flush=True)
torch.cuda.empty_cache()
print("after mixed + empty_cache:", bench.angular_config3(ctx, dev, torch, np)["compute_ms"], flush=True)
m = bench.mixed_share(ctx, dev, torch, np)
print("after mixed, no empty_cache:", bench.angular_config3(ctx, dev, torch, np)["compute_ms"], flush=True)
torch.cuda.empty_cache()
print("then empty_cache:", bench.angular_config3(ctx, dev, torch, np)["compute_ms"], flush=True)
print(torch.cuda.memory_summary(abbreviated=True)[:1500])
