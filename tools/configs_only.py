import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, bench, json
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
M = bench.load_blosum62()
out = bench.other_configs(M, dev, torch.cuda.current_stream(dev))
for c in out: print(c["config"][:40], round(c["kernel_ms"],4), round(c["roofline"]["frac"],3))
