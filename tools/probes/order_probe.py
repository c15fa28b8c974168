import sys, numpy as np
sys.path.insert(0, "/root/repo")
import __graft_entry__ as e; e.build()
from covid19uk_amd import synth
from covid19uk_amd.seir import SeirModel
cov = synth.make_covariates("ni11"); ev, init, tr = synth.simulate_epidemic(cov)
with SeirModel(cov, init) as m:
    pass
import torch
print("after SeirModel: is_available", torch.cuda.is_available(), "count", torch.cuda.device_count())
