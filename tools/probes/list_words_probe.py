"""diagnostics: mean words per tile of the list loop on the bench workload, lattice start vs melted (padding of the tiles)"""
import importlib, sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
inp = importlib.import_module("ls1-mardyn_amd.inp"); engine_mod = importlib.import_module("ls1-mardyn_amd.engine"); synth = importlib.import_module("ls1-mardyn_amd.synth")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 171
comps = inp.ComponentSet([inp.make_component(lj=[(0, 0, 0, 1, 1, 1, 2.5, 0)])], np.zeros((0, 2)), 1e10)
L = synth.box_length(n)
e = engine_mod.DeviceEngine(0); e.set_components(comps, 2.5); e.set_verlet(0.2); e.set_domain([L] * 3)
N = 2 * n ** 3
e.upload_begin(N)
for ids_t, r_t, v_t in synth.bcc_chunks_device(torch, torch.device("cuda", 0), n):
    torch.cuda.synchronize(); e.upload_chunk_device(ids_t.numel(), ids_t.data_ptr(), 0, r_t.data_ptr(), v_t.data_ptr())
e.upload_end(); e.rebin(); e.halo(); e.forces(0)
rho = N / L ** 3
print("ideal entries within rc+skin in a liquid:", 4 / 3 * np.pi * 2.7 ** 3 * rho)
for steps in (1, 60, 200, 400):
    e.run(0.002, steps)
    print("after", steps, "more steps: builds", e.get_option("verlet_builds"), "mean words/tile", e.get_option("verlet_mean_words_x1000") / 1000.,
          "= entries walked per molecule", 4 * e.get_option("verlet_mean_words_x1000") / 1000.)
