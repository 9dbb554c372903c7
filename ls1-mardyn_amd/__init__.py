"""ls1-mardyn_amd — MI355X-native linked-cell pair-force engine behind the ls1-MarDyn plug-in API.

Layout: csrc/ (HIP kernels + C ABI -> lib/libls1hip.so), capi.py (ctypes binding), engine.py (context owner),
mirror.py (host-side mirror of the reference's ParticleContainer / CellProcessor / Integrator / DomainDecompBase
interfaces for this path), inp.py (.inp phase-space reader), decomp.py (multi-GPU domain decomposition over
torch.distributed / RCCL).  The package directory name contains a hyphen: import it with
``importlib.import_module("ls1-mardyn_amd")``.
"""
__all__ = ["capi", "engine", "inp", "mirror", "decomp"]
