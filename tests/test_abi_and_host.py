"""CPU tests: the C-ABI library loads and exports every symbol include/lcp_hip.h declares (no
compute calls without a GPU), host-side logic (workspace sizing, argument checks, FLOP counts,
scene generators) and the no-fallback rule."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "lcp_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(lcp_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_all_exported_and_bound():
    from lcp_physics_amd import _lib
    lib = _lib.load()
    syms = _declared_symbols()
    assert "lcp_pdipm_forward_f32" in syms and "lcp_step_fused_f32" in syms and len(syms) >= 8
    for s in syms:
        assert hasattr(lib, s), "missing export %s" % s
        assert s in _lib.SIGNATURES, "no ctypes signature for %s" % s
    assert lib.lcp_version().decode().startswith("lcp_hip")


def test_workspace_bytes_and_argument_checks_host_only():
    from lcp_physics_amd import _lib
    lib = _lib.load()
    w64 = lib.lcp_workspace_bytes(4096, 15, 64, 3, _lib.COMPUTE_F64)
    w32 = lib.lcp_workspace_bytes(4096, 15, 64, 3, _lib.COMPUTE_F32)
    assert w64 > w32 and w64 >= 4096 * 8 * (64 * 64 + 15 * 15 + 64 * 3 + 9 + 15 + 128 + 3)
    tail = ((4096 * 4 + 255) & ~255) + 256                      # behind the scene blocks: per-scene classes of the dense lcp_big path + the layout-tag trailer
    assert (w64 - tail) % 4096 == 0 and ((w64 - tail) // 4096) % 256 == 0        # per-scene stride keeps 256 B alignment
    # fp64 I/O keeps an fp64 copy of F in the workspace: the caller says so with LCP_IO_F64
    assert lib.lcp_workspace_bytes(4096, 15, 64, 3, _lib.COMPUTE_F64 | _lib.IO_F64) > w64
    # config-5 sizes (nz 33, nineq 256): lcp_big.hip's layout (W tiles + iterate) or the generic one, whichever is larger
    assert lib.lcp_workspace_bytes(8, 33, 256, 3, _lib.COMPUTE_F64) >= 8 * 8 * (4 * 64 * 64 + 64 + 72 + 10 * 64)
    assert lib.lcp_workspace_bytes(0, 15, 64, 3, 1) == 0
    # argument validation happens before any launch: NULL pointers / bad sizes -> LCP_E_BADARG
    N = None
    rc = lib.lcp_pdipm_forward_f32(4, 15, 64, 3, N, N, N, N, N, N, N, 1e-12, 10, 3, 1, N, N, N, N, N, N, N, N)
    assert rc == -1
    rc = lib.lcp_pdipm_forward_f32(0, 15, 64, 3, N, N, N, N, N, N, N, 1e-12, 10, 3, 1, N, N, N, N, N, N, N, N)
    assert rc == -1
    rc = lib.lcp_pdipm_backward_f32(4, 15, 64, 3, N, N, N, 1, N, N, N, N, N, N, N, N, N)
    assert rc == -1
    # which contact-list sizes have a fused backward (host-only planning query; the family is a function of sizes + compute word)
    assert lib.lcp_step_has_backward(5, 16, 3, _lib.COMPUTE_F64) == 1            # four scenes per wave
    assert lib.lcp_step_has_backward(11, 64, 3, _lib.COMPUTE_F64) == 1           # body space, one wave per scene
    assert lib.lcp_step_has_backward(11, 40, 20, _lib.COMPUTE_F64) == 1          # chains of joints
    assert lib.lcp_step_has_backward(5, 16, 3, _lib.COMPUTE_F64 | _lib.PATH_GENERIC) == 1   # (round 6) the generic kernels keep their iterate too
    assert lib.lcp_step_has_backward(20, 64, 3, _lib.COMPUTE_F64) == 1           # 3 nb + e = 63: the 64-row body-space kernel (round 6)
    assert lib.lcp_step_has_backward(21, 64, 3, _lib.COMPUTE_F64) == 1           # 66 rows: lcp_step_kernel + lcp_step_bwd_kernel (generic)
    assert lib.lcp_step_has_backward(40, 128, 3, _lib.COMPUTE_F64) == 1          # beyond 64 contacts
    assert lib.lcp_step_has_backward(11, 64, 3, _lib.COMPUTE_F32) == 1
    assert lib.lcp_step_has_backward(5, 16, 6, _lib.COMPUTE_F32) == 0            # the wave64 step family (fp32, 5..8 joint rows): forward only
    assert lib.lcp_step_has_backward(100, 300, 3, _lib.COMPUTE_F64) == 0         # beyond the generic plan (LCP_E_TOOLARGE forward as well)


def test_no_cpu_fallback():
    """The op must refuse CPU-only execution instead of silently computing elsewhere."""
    from lcp_physics_amd import scenes
    from lcp_physics_amd.lcp import LCPFunction, lcp_solve
    lcp = scenes.make_random_lcp(2, 5, 8, 0, dtype=torch.float32)
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError, match="GPU"):
        lcp_solve(lcp[0], lcp[1], lcp[2], lcp[3], None, None, lcp[6])
    with pytest.raises(RuntimeError, match="GPU"):
        LCPFunction()(lcp[0], lcp[1], lcp[2], lcp[3], torch.tensor([]), torch.tensor([]), lcp[6])


def test_product_package_never_imports_oracle():
    pkg = os.path.join(ROOT, "lcp_physics_amd")
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h")):
                txt = open(os.path.join(d, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt, os.path.join(d, f)


def test_flop_model_matches_survey_figures():
    from lcp_physics_amd import flops
    assert abs(flops.flops_forward(15, 64, 3, 10) / 1e6 - 2.56) < 0.01      # SURVEY §8(d), cfg 3
    assert abs(flops.flops_backward(15, 64, 3) / 1e6 - 0.198) < 0.002
    assert flops.bytes_forward(15, 64, 3) == 22216
    assert flops.bytes_forward(15, 64, 3) + flops.bytes_backward(15, 64, 3) == 65796
    assert abs(flops.flops_forward(9, 32, 3, 10) / 1e6 - 0.41) < 0.01       # cfg 2


def test_stack_scene_generator_shapes_and_conventions():
    from lcp_physics_amd import scenes
    from oracle import pdipm_oracle as O
    sc = scenes.make_stack_scenes(B=5, nbox=4, pts_per_interface=4, seed=1)
    assert sc.v.shape == (5, 5, 3) and sc.c_n.shape == (5, 16, 2) and sc.Je.shape == (5, 3, 15)
    assert sc.c_i1.dtype == torch.int32 and int(sc.c_i1.min()) == 0 and int(sc.c_i2.max()) == 4
    assert bool((sc.c_i2 == sc.c_i1 + 1).all())
    # floor/box contact rows look like the reference's: n = [0, 1], p1.y = -5, p2.y = 30 + gap
    assert torch.allclose(sc.c_p1[:, :4, 1], torch.full((5, 4), -5.0))
    assert torch.allclose(sc.c_p2[:, :4, 1], torch.full((5, 4), 30.0 + scenes.GAP))
    Q, p, G, h, A, b, F = O.assemble_lcp(*sc.assembly_args())
    assert G.shape == (5, 64, 15) and F.shape == (5, 64, 64) and A.shape == (5, 3, 15)
    assert torch.equal(G[:, 48:], torch.zeros(5, 16, 15))                     # gamma rows of G are zero
    assert torch.allclose(G[:, 16:48:2], -G[:, 17:48:2])                     # +/- friction directions
    sc2 = scenes.make_stack_scenes(B=5, nbox=4, pts_per_interface=4, seed=1)
    assert torch.equal(sc.v, sc2.v) and torch.equal(sc.c_p1, sc2.c_p1)        # seeded
    pile = scenes.make_pile_scenes(B=2)
    assert pile.nb == 11 and pile.nc == 64


def test_pure_fp32_reference_style_misses_tolerance_documented():
    """Documents WHY the parity path computes in fp64: the reference algorithm run in fp32 (with the
    CPU's partial pivoting) misses 1e-4 on a visible fraction of scenes (numerics)."""
    from lcp_physics_amd import scenes
    from oracle import pdipm_oracle as O
    from tests import parity
    sc = scenes.make_stack_scenes(B=128, nbox=4, pts_per_interface=4, seed=1236, dtype=torch.float32)
    lcp32 = O.assemble_lcp(*sc.assembly_args())
    lcp64 = [None if t is None else t.double() for t in lcp32]
    ref = O.lcp_forward(*lcp64)
    s32 = O.lcp_forward(*lcp32)
    ex = parity.err_x(s32.x.double(), ref.x, lcp64[0], lcp64[1])
    assert float(ex.median()) < 1e-5          # typical scenes are fine ...
    assert float(ex.max()) > 1e-4             # ... but the tail is not


def test_geometry_batch_matches_reference_rect_convention():
    """Host side of the contact path (no GPU): `GeometryBatch.from_shapes` stores a Rect's body-frame vertices in the
    order of the reference (`bodies.py:261-264`: v0 = half, v1 = half * (-1, 1), -v0, -v1), which is what
    oracle.contacts_oracle.rect_verts produces at rotation 0; circles carry a radius and no vertices."""
    import numpy as np
    from lcp_physics_amd.physics.contacts import CIRCLE, HULL, NV, GeometryBatch
    from oracle import contacts_oracle as C
    tri = [[0.0, -2.0], [2.0, 1.0], [-2.0, 1.0]]
    g = GeometryBatch.from_shapes([("rect", (4.0, 2.0)), ("circle", 1.5), ("hull", tri)], B=3)
    assert tuple(g.verts_local.shape) == (3, 3, NV, 2) and g.kind[0].tolist() == [HULL, CIRCLE, HULL]
    assert np.allclose(g.verts_local[1, 0, :4].numpy(), C.rect_verts((4.0, 2.0), 0.0))
    assert g.nverts[2].tolist() == [4, 0, 3] and float(g.radius[0, 1]) == 1.5
    assert np.allclose(g.verts_local[0, 2, :3].numpy(), np.array(tri)) and float(g.verts_local[0, 2, 3:].abs().max()) == 0.0
    with pytest.raises(ValueError):
        GeometryBatch.from_shapes([("hull", [[float(i), float(i * i)] for i in range(NV + 1)])])


def test_contact_path_refuses_cpu_tensors():
    """No CPU fallback on the contact path either."""
    from lcp_physics_amd.physics.contacts import GeometryBatch, find_contacts
    g = GeometryBatch.from_shapes([("rect", (4.0, 2.0)), ("circle", 1.5)], B=1)
    with pytest.raises(RuntimeError):
        find_contacts(g, torch.zeros(1, 2, 3, dtype=torch.float64))
