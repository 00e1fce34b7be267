import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import qrkit_amd
import bench
ctx = qrkit_amd.Context(0)
dev = torch.device("cuda", 0)
print("alone:", bench.angular_config3(ctx, dev, torch, np)["compute_ms"], flush=True)
m = bench.mixed_share(ctx, dev, torch, np)
print("mixed:", m["ms_best"], flush=True)
print("after mixed:", bench.angular_config3(ctx, dev, torch, np)["compute_ms"], flush=True)
s = bench.strips_config2(ctx, dev, torch, np, 2048)
print("strips:", s["ms_per_strip"], flush=True)
print("after strips 2048:", bench.angular_config3(ctx, dev, torch, np)["compute_ms"], flush=True)
time.sleep(3)
print("after 3 s idle:", bench.angular_config3(ctx, dev, torch, np)["compute_ms"], flush=True)
ctx2 = qrkit_amd.Context(0)
print("fresh context:", bench.angular_config3(ctx2, dev, torch, np)["compute_ms"], flush=True)
