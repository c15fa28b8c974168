import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: larger CPU-only case")


import warnings

import numpy as np

# divergent HMC trajectories legitimately overflow inside the oracle (-> NaN -> reject)
warnings.filterwarnings("ignore", category=RuntimeWarning, module=r"oracle\..*")
np.seterr(all="ignore")
