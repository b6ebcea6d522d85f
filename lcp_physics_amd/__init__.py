"""MI355X-native batched LCP contact solver: drop-in for lcp-physics' `LCPFunction`
(`lcp_physics/lcp/lcp.py`) and `Engine.solve_dynamics` (`lcp_physics/physics/engines.py`).

The compute path is hand-written HIP for gfx950 behind a C ABI (`include/lcp_hip.h`,
`lcp_physics_amd/csrc/`); there is no CPU fallback - using an op without the built
library raises.
"""
__version__ = "0.1.0"
