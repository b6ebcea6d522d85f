from .engines import Engine, HipPdipmEngine, HipFusedEngine  # noqa: F401
from .batched_world import BatchedWorld, fused_step, assemble_contacts  # noqa: F401
