import sys, os, time, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import bench
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
for i in range(3):
    out = bench.other_configs(dev)
    print({k: round(v["ms"], 3) for k, v in out.items()})
