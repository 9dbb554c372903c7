"""Worker for tests/test_decomp_cpu.py: one rank of a gloo world running decomp.DistributedSimulation over the
numpy stand-in engine; rank 0 gathers the final state and writes it to an .npz."""
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
    decomp = importlib.import_module("ls1-mardyn_amd.decomp")
    from cpu_engine import CpuEngine
    data = np.load(os.environ["LS1_TEST_INPUT"])
    L, r, v, ids, rc = data["L"], data["r"], data["v"], data["ids"], float(data["rc"])
    dc = decomp.CartesianDecomposition(world, rank, L, grid, loopback=bool(int(os.environ.get("LS1_TEST_LOOPBACK", "0"))))
    lo, hi = dc.bounding_box()
    mine = np.all((r >= lo) & (r < hi), axis=1)
    skin = float(os.environ["LS1_TEST_SKIN"]) if os.environ.get("LS1_TEST_SKIN") else None
    eng = CpuEngine(L, lo, hi, rank, dc.neighbor_table(), rc, skin=skin)
    eng.upload(ids[mine], r[mine], v[mine])
    fail = os.environ.get("LS1_TEST_FAIL", "")  # "<rank>:<export_counts call number>:<export|import>"
    if fail and int(fail.split(":")[0]) == rank:
        at, where = int(fail.split(":")[1]), fail.split(":")[2]
        name = "export_counts" if where == "export" else "import_done"
        orig, calls = getattr(eng, name), [0]

        def failing(*a, **k):
            calls[0] += 1
            if calls[0] == at:
                raise RuntimeError(f"injected engine failure in {name} on rank {rank}")
            return orig(*a, **k)
        setattr(eng, name, failing)
    sim = decomp.DistributedSimulation(dc, eng, dist, torch.device("cpu"))
    macro0 = sim.initial_forces()
    g0 = sim.reduce_globals(macro0, (0.0, 0.0, len(eng.ids), 0))
    F0 = eng.F.copy(); ids0 = eng.ids.copy()
    res = sim.run(dt, nsteps) if nsteps else g0
    payload = dict(ids=eng.ids, r=eng.r, v=eng.v, F=eng.F, ids0=ids0, F0=F0)
    gathered = [None] * world
    dist.gather_object(payload, gathered if rank == 0 else None, dst=0)
    if rank == 0:
        cat = {k: np.concatenate([g[k] for g in gathered]) for k in payload}
        np.savez(out_path, upot0=g0["upot"], virial0=g0["virial"], upot=res["upot"], virial=res["virial"],
                 summv2=res["summv2"], n=res["n"], **cat)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
