"""MI355X-native 1-D electrostatic PIC stepper behind the reference's `PIC` environment surface.

The directory name is fixed by the project layout and is not an importable identifier; import
it through the repo-root alias::

    import ocplasma_amd
    from ocplasma_amd.env.pic import PIC          # drop-in for src/env/pic.py
    from ocplasma_amd.env.dist import BumpOnTail, TwoStream
    from ocplasma_amd.control.actuator import E_field
    from ocplasma_amd.control.reward import Reward

Everything that steps particles goes through csrc/libpicstep.so (HIP, gfx950); there is no CPU
fallback.
"""
from . import _abi, _build
from .env import PIC, BatchedPIC, ShardedPIC, TwoStream, BumpOnTail
from .control import E_field, Reward
from .interpret import compute_E_k_spectrum

__all__ = ["PIC", "BatchedPIC", "ShardedPIC", "TwoStream", "BumpOnTail", "E_field", "Reward", "compute_E_k_spectrum"]
