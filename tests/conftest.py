import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The product refuses to import without its in-tree library.  On a checkout that has not been built yet
    # (the .so files are not in git) build it the way __graft_entry__.build() does, before test modules import it.
    lib = os.path.join(ROOT, "hammock_amd", "lib", "libhammock_hip.so")
    cli = os.path.join(ROOT, "hammock_amd", "bin", "hammock-hip")
    if not (os.path.exists(lib) and os.path.exists(cli)):
        import subprocess
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "hammock_amd", "csrc"), "-j4"], stdout=subprocess.DEVNULL)


@pytest.fixture(scope="session")
def matrices():
    with open(os.path.join(GOLDEN, "matrices.json")) as fh:
        d = json.load(fh)
    return {k: np.asarray(v, dtype=np.int32) for k, v in d["matrices"].items()}


@pytest.fixture(scope="session")
def blosum62(matrices):
    return matrices["blosum62"]


@pytest.fixture(scope="session")
def known_answers():
    with open(os.path.join(GOLDEN, "known_answers.json")) as fh:
        return json.load(fh)


@pytest.fixture(scope="session")
def coracle():
    from oracle import c_oracle
    c_oracle.lib()
    return c_oracle


def gpu_count():
    """GPUs visible to this process (counting devices does not initialise HIP)."""
    try:
        import torch
        return int(torch.cuda.device_count())
    except Exception:
        return 0


def multi_device_lists(repeated=((0, 0), (0, 0, 0))):
    """Device lists for the hmk_create_multi tests: the one GPU named several times (separate contexts, plans, buffers and
    streams on one card; runs everywhere) AND, on a box with more GPUs, distinct ordinals -- [0, 1], [1, 0] and all of them --
    which is what exercises hipDeviceEnablePeerAccess and real peer copies.  The distinct lists are always generated and are
    skipped, visibly, where fewer than two GPUs exist: nobody has to edit a test the day a multi-GPU box runs them."""
    n = gpu_count()
    out = [pytest.param(list(d), id="dev" + "".join(map(str, d))) for d in repeated]
    need2 = pytest.mark.skipif(n < 2, reason="needs two GPUs (distinct ordinals: peer access, peer copies over xGMI)")
    out.append(pytest.param([0, 1], id="dev01_distinct", marks=need2))
    out.append(pytest.param([1, 0], id="dev10_distinct", marks=need2))
    out.append(pytest.param(list(range(max(n, 3))), id="dev_all_distinct",
                            marks=pytest.mark.skipif(n < 3, reason="needs three or more GPUs")))
    return out


def random_peptides(rng, n, len_lo, len_hi, alphabet=20):
    """n DISTINCT random peptides as a list of uint8 arrays."""
    seen, out = set(), []
    while len(out) < n:
        L = int(rng.integers(len_lo, len_hi + 1))
        p = rng.integers(0, alphabet, size=L, dtype=np.uint8)
        key = p.tobytes()
        if key in seen:
            continue
        seen.add(key)
        out.append(p)
    return out


def hand_traces():
    """tests/golden/hand_traces.json + the matrix it describes (int32 [24][24])."""
    import json
    import numpy as np
    with open(os.path.join(GOLDEN, "hand_traces.json")) as fh:
        ht = json.load(fh)
    alphabet = "ARNDCQEGHILKMFPSTWYVBZX*"
    m = ht["matrix"]
    M = np.full((24, 24), m["off_diagonal"], dtype=np.int32)
    for k, a in enumerate(alphabet):
        M[k, k] = m["diagonal"].get(a, m["diagonal_default"])
    return ht, M
