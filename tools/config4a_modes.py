import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))) if '__file__' in dir() else '.')
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import hammock_amd
from hammock_amd import _native
from hammock_amd.synth import synth_peptides
from bench import load_blosum62
res, off = synth_peptides(1, 100000, 7, 20)
ctx = hammock_amd.Context(load_blosum62(), device=0)
ctx.set_sequences(residues=res, offsets=off)
dev = torch.device("cuda", 0)
cap = int(os.environ.get("CAP", str(1 << 24)))
d_edges = torch.empty(cap, dtype=torch.int64, device=dev)
d_counts = torch.zeros(_native.HMK_EDGE_SHARDS, dtype=torch.int64, device=dev)
stream = torch.cuda.current_stream(dev)
def one(): ctx.neighbors_shifted_dev(3, -1, 23, 0, 1, d_edges.data_ptr(), cap, d_counts.data_ptr(), stream.cuda_stream)
for _ in range(10): one()
torch.cuda.synchronize()
# per-pass sync
ms=[]
for _ in range(16):
    a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    a.record(stream); one(); b.record(stream); torch.cuda.synchronize(); ms.append(a.elapsed_time(b))
print("sync per pass: median", round(float(np.median(ms)),3), "min", round(min(ms),3))
# back to back with events
evs=[(torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)) for _ in range(16)]
t=time.perf_counter()
for a,b in evs:
    a.record(stream); one(); b.record(stream)
torch.cuda.synchronize()
wall=(time.perf_counter()-t)/16*1e3
ms=[a.elapsed_time(b) for a,b in evs]
print("back to back: median", round(float(np.median(ms)),3), "min", round(min(ms),3), "wall per pass", round(wall,3))
