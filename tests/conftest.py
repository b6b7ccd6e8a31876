import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the pooled-build tests move device buffers with torch: torch's HIP runtime has to be the one that initialises the GPU
    # (genomeassembler_dev_amd/_lib.py), so it is imported before any test loads libgasm
    if "gpu" in (config.getoption("-m") or "") and "not gpu" not in (config.getoption("-m") or ""):
        try:
            import torch
            if torch.cuda.is_available():
                torch.cuda.init()
        except Exception:
            pass


@pytest.fixture(scope="session")
def qtable():
    """(keys, normalised probs) of the 69 904-row breakage table, A15 normalisation by the ORACLE."""
    import itertools

    import numpy as np

    from oracle import orc
    raw = np.fromfile(os.path.join(ROOT, "genomeassembler_dev_amd", "data", "querytable_raw_f64.bin"), dtype="<f8")
    prob = orc.normalise_tables(raw, [16, 256, 4096, 65536])
    keys = ["".join(t) for k in (2, 4, 6, 8) for t in itertools.product("ACGT", repeat=k)]
    return keys, prob
