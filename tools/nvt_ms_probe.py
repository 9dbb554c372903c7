"""NVT list loop of the 2CLJ ethane box (replicated k^3) on the device: ms per step of ls1hip_run with the velocity-scaling thermostat at the
kinetic temperature of the fixture's state.  usage: python tools/nvt_ms_probe.py [k = 10] [steps = 60]   (LS1HIP_LIB selects the library)"""
import importlib, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
inp = importlib.import_module("ls1-mardyn_amd.inp")
engine_mod = importlib.import_module("ls1-mardyn_amd.engine")
k = int(sys.argv[1]) if len(sys.argv) > 1 else 10
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
ps0, _ = bench.ethane_fixture(inp)
big = bench.replicate_phase_space(inp, ps0, k)
N = len(big.ids)
e = engine_mod.DeviceEngine(0)
e.set_components(big.components, bench.ETHANE_RC)
e.set_verlet(6.0)
e.set_domain(big.length)
e.set_thermostat(True, 2.3755e-4)
e.upload(big.ids, big.cid, big.r, big.v, big.q, big.D)
e.rebin(); e.halo(); e.forces(0)
e.run(bench.ETHANE_DT, 10)
e.synchronize() if hasattr(e, "synchronize") else None
t0 = time.time()
out = e.run(bench.ETHANE_DT, steps)
wall = time.time() - t0
print(f"ethane NVT N={N} steps={steps}: {wall/steps*1e3:.3f} ms/step ({N*steps/wall:.4g} updates/s), list_kick_available={e.get_option('list_kick_available')}, "
      f"builds {e.get_option('verlet_builds')}, T_kin={(out['summv2']+out['sumIw2'])/(5*N):.6e}")
