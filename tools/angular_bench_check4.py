import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import qrkit_amd
import bench
ctx = qrkit_amd.Context(0)
dev = torch.device("cuda", 0)
tag = " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith(("QRK_CAQR", "GPU_MAX")))
a = bench.angular_config3(ctx, dev, torch, np)["compute_ms"]
m = bench.mixed_share(ctx, dev, torch, np)
b = bench.angular_config3(ctx, dev, torch, np)["compute_ms"]
print(f"[{tag}] alone {a:.1f} ms, after the mixed batch {b:.1f} ms", flush=True)
