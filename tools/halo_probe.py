"""Times the halo phase alone (rebin once, then repeated ls1hip_halo) on the bench workload."""
import importlib, sys, os
import numpy as np
sys.path.insert(0, os.getcwd())
import bench
inp = importlib.import_module("ls1-mardyn_amd.inp")
capi = importlib.import_module("ls1-mardyn_amd.capi")
if len(sys.argv) > 1:
    capi.LIB_PATH = os.path.abspath(sys.argv[1])  # probe a library variant
engine_mod = importlib.import_module("ls1-mardyn_amd.engine")
import torch
L, _ids, r, v = __import__("importlib").import_module("ls1-mardyn_amd.synth").bcc_box(171)
eng = engine_mod.DeviceEngine(0)
eng.set_components(bench.lj_components(inp), bench.RC)
eng.set_domain([L, L, L])
N = len(r)
eng.upload(np.arange(1, N + 1, dtype=np.uint64), np.zeros(N, np.int32), r, v)
eng.rebin(); eng.halo()
eng.timing_reset(); eng.timing_enable(True)
for _ in range(20):
    eng.halo()
torch.cuda.synchronize()
ms, n = eng.timing("halo")
print("halo phase ms:", ms / n)
