"""Host-side latency of the drop-in engine at batch ONE - the only way the reference's `physics` layer ever calls its engine
(engines.py:54-76).  The reference's `World` cannot travel to the GPU box, its recorded accessor answers can
(tests/world_io.py::RecordedWorld over tests/golden/steps_*.npz): `HipPdipmEngine.solve_dynamics(world, dt)` is timed per recorded step,
warm (a first call pays the one-time load of the code object), differentiable and not, and the share of the call that is `_Lifted`
(host packing + the two transfers).  Prints one JSON object.   python tools/experiments/engine_latency.py"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import golden_io                      # noqa: E402
from tests.world_io import RecordedWorld          # noqa: E402


def main():
    from lcp_physics_amd.physics import HipFusedEngine, HipPdipmEngine
    from lcp_physics_amd.physics.engines import _Lifted
    out = {"what": "ms per engine.solve_dynamics(world, dt) call at batch 1 on recorded reference worlds (warm; median of the repeats)", "scenes": {}}
    for name in golden_io.SCENES:
        steps = golden_io.load_steps(name)
        st = steps[len(steps) // 2]
        rec = {"bodies": int(st["v"].shape[0]), "contacts": int(st["c_n"].shape[0])}
        for label, Eng, leaf in (("fused", HipFusedEngine, False), ("differentiable", HipPdipmEngine, True)):
            eng = Eng()
            world = RecordedWorld(st, leaf=leaf)
            for _ in range(5):
                eng.solve_dynamics(world, st["dt"])
            ts = []
            for _ in range(50):
                t0 = time.perf_counter()
                v = eng.solve_dynamics(world, st["dt"])
                ts.append(time.perf_counter() - t0)
            ts.sort()
            rec["ms_" + label] = ts[len(ts) // 2] * 1e3
        world = RecordedWorld(st)
        ts = []
        for _ in range(50):
            t0 = time.perf_counter()
            _Lifted(world)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        ts.sort()
        rec["ms_lift_only"] = ts[len(ts) // 2] * 1e3
        out["scenes"][name] = rec
    print(json.dumps(out))


if __name__ == "__main__":
    main()
