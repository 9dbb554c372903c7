"""Worker for tests/test_gpu_multiproc.py: one rank of a multi-PROCESS world in which every rank drives its own DeviceEngine
context through decomp.DistributedSimulation.  All ranks share cuda:0 (RCCL wants one GPU per rank, so the transport is gloo
with the messages staged through the host: HaloExchanger(stage_through_host=True)); everything else — export / pack /
import kernels, stream order, the collective rebuild decision, the reductions — is the code a multi-GPU run executes.
rank 0 gathers the state and writes it to an .npz."""
import importlib
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)


def main():
    out_path, nsteps, dt = sys.argv[1], int(sys.argv[2]), float(sys.argv[3])
    grid = tuple(int(x) for x in sys.argv[4].split("x")) if len(sys.argv) > 4 else None
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    decomp = importlib.import_module("ls1-mardyn_amd.decomp")
    engine_mod = importlib.import_module("ls1-mardyn_amd.engine")
    inp = importlib.import_module("ls1-mardyn_amd.inp")
    data = np.load(os.environ["LS1_TEST_INPUT"])
    L, r, v, ids, rc = data["L"], data["r"], data["v"], data["ids"], float(data["rc"])
    skin = float(os.environ["LS1_TEST_SKIN"]) if os.environ.get("LS1_TEST_SKIN") else None
    fuse = os.environ.get("LS1_TEST_FUSE", "1") == "1"
    comps = inp.ComponentSet([inp.make_component(lj=[(0, 0, 0, 1, 1, 1, rc, 0)])], np.zeros((0, 2)), 1e10)
    dc = decomp.CartesianDecomposition(world, rank, L, grid)
    lo, hi = dc.bounding_box()
    mine = np.all((r >= lo) & (r < hi), axis=1)
    eng = engine_mod.DeviceEngine(0)
    eng.set_components(comps, rc)
    if skin:
        eng.set_verlet(skin, force=True)
    eng.set_domain(L, lo, hi, rank, dc.neighbor_table())
    eng.upload(ids[mine], np.zeros(int(mine.sum()), np.int32), r[mine], v[mine])
    sim = decomp.DistributedSimulation(dc, eng, dist, torch.device("cuda", 0), stage_through_host=True)
    macro0 = sim.initial_forces()
    g0 = sim.reduce_globals(macro0, (0.0, 0.0, eng.count()[0], 0))
    F0 = eng.download_forces()["F"].copy()
    ids0 = eng.download_ids().copy()
    res = sim.run(dt, nsteps, fuse=fuse) if nsteps else g0
    st = eng.download_state()
    payload = dict(ids=st["ids"], r=st["r"], v=st["v"], F=eng.download_forces()["F"], ids0=ids0, F0=F0)
    gathered = [None] * world
    dist.gather_object(payload, gathered if rank == 0 else None, dst=0)
    if rank == 0:
        cat = {k: np.concatenate([g[k] for g in gathered]) for k in payload}
        np.savez(out_path, upot0=g0["upot"], virial0=g0["virial"], upot=res["upot"], virial=res["virial"],
                 summv2=res["summv2"], n=res["n"], **cat)
    eng.close()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
