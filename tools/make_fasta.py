import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from hammock_amd.synth import synth_peptides
n=int(sys.argv[1])
res, off = synth_peptides(1, n, 12)
A=np.frombuffer(b"ARNDCQEGHILKMFPSTWYV", dtype=np.uint8)
seq=A[res].reshape(n,12)
with open(sys.argv[2],"wb") as f:
    lines=[]
    for i in range(n):
        lines.append(b">%d\n%s\n" % (i, seq[i].tobytes()))
    f.write(b"".join(lines))
