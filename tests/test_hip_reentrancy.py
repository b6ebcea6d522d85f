"""GPU: SURVEY.md §8b "Threading" - the C ABI holds no process-global mutable state (host threads may drive different
streams concurrently, the lcp_debug_* settings are per thread), and a truncated contact list is LOUD (status bit, sticky
flag in ContactWorld) instead of a silent clamp."""
import threading

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _scenes(seed, B=512, nbox=4):
    from lcp_physics_amd import scenes
    return scenes.make_stack_scenes(B=B, nbox=nbox, pts_per_interface=4, seed=seed, dtype=torch.float32).to(device=DEV)


def _run(sc, path_setting, reps, stream=None, results=None, key=None):
    """`reps` fused steps + dense backward on `stream`, kernel family forced through the PER-THREAD debug setting."""
    from lcp_physics_amd import _lib
    from lcp_physics_amd.lcp import lcp_backward
    from lcp_physics_amd.physics import assemble_contacts, fused_step
    from lcp_physics_amd.physics.batched_world import solution_of_step
    _lib.set_path(path_setting)
    try:
        with torch.cuda.stream(stream) if stream is not None else torch.cuda.stream(torch.cuda.current_stream()):
            lcp = assemble_contacts(sc)
            cot = torch.ones(sc.B, 3 * sc.nb, device=DEV)
            out = None
            for _ in range(reps):
                out = fused_step(sc, ws=None if out is None else out["ws"], out=out)
                grads = lcp_backward(solution_of_step(sc, out, lcp[2], lcp[4]), cot)
            if stream is not None:
                stream.synchronize()
            else:
                torch.cuda.synchronize()
            res = {k: out[k].clone() for k in ("v_new", "p_new", "z", "s", "iters", "status")}
            res["dp"] = grads[1].clone()
            res["dG"] = grads[2].clone()
    finally:
        _lib.set_path("auto")
    if results is not None:
        results[key] = res
    return res


def test_two_threads_two_streams_two_path_settings_match_the_serial_runs():
    sa, sb = _scenes(11), _scenes(12)
    serial_a = _run(sa, "auto", 3)
    serial_b = _run(sb, "generic", 3)
    # the two settings really select different kernels (otherwise the test proves nothing): same answers to rounding only
    cross = _run(sa, "generic", 1)
    assert not torch.equal(cross["z"], serial_a["z"]) and torch.allclose(cross["v_new"], serial_a["v_new"], atol=1e-5)
    results = {}
    st_a, st_b = torch.cuda.Stream(), torch.cuda.Stream()
    ta = threading.Thread(target=_run, args=(sa, "auto", 3, st_a, results, "a"))
    tb = threading.Thread(target=_run, args=(sb, "generic", 3, st_b, results, "b"))
    ta.start(); tb.start(); ta.join(); tb.join()
    assert set(results) == {"a", "b"}
    for k in serial_a:
        assert torch.equal(results["a"][k], serial_a[k]), ("thread a", k)
        assert torch.equal(results["b"][k], serial_b[k]), ("thread b", k)


def test_path_setting_of_one_thread_does_not_leak_into_another():
    from lcp_physics_amd import _lib
    sa = _scenes(13, B=64)
    ref_auto = _run(sa, "auto", 1)
    _lib.set_path("generic")                       # this (main) thread only
    try:
        results = {}
        t = threading.Thread(target=lambda: results.__setitem__("t", _run_no_setting(sa)))
        t.start(); t.join()
    finally:
        _lib.set_path("auto")
    assert torch.equal(results["t"]["z"], ref_auto["z"])           # the worker thread saw the default (auto), not "generic"


def _run_no_setting(sc):
    from lcp_physics_amd.physics import fused_step
    out = fused_step(sc)
    torch.cuda.synchronize()
    return {"z": out["z"].clone()}


@pytest.mark.parametrize("nbox,maxc", [(4, 8), (6, 12), (6, 20)])
def test_truncated_contact_list_raises_the_status_bit(nbox, maxc):
    """count > maxc: every solver family reports LCP_ST_TRUNCATED for exactly those scenes (quad, quad-wide, big classes)."""
    from lcp_physics_amd import _lib, scenes
    from lcp_physics_amd.physics.batched_world import solve_dynamics
    from lcp_physics_amd.physics.contacts import ContactBuffers
    B = 16
    sc = scenes.make_stack_scenes(B=B, nbox=nbox, pts_per_interface=4, seed=3, dtype=torch.float32).to(device=DEV)
    assert sc.nc > maxc
    cb = ContactBuffers(B, sc.nb, maxc, DEV)
    for k in ("c_n", "c_p1", "c_p2", "c_i1", "c_i2"):
        setattr(cb, k, getattr(sc, k)[:, :maxc].contiguous())
    count = torch.full((B,), maxc, dtype=torch.int32, device=DEV)
    count[::3] = sc.nc                                               # these scenes "found" more contacts than the list holds
    for path in ("auto", "generic"):
        out = solve_dynamics(B, sc.nb, maxc, 3, count, sc.Mdiag, sc.v, sc.f, sc.rest, sc.fric, cb, sc.Je, sc.dt, path=path)
        st = out["status"].cpu()
        assert bool(((st & _lib.ST_TRUNCATED) != 0)[::3].all()), path
        rest = torch.ones(B, dtype=torch.bool); rest[::3] = False
        assert not bool(((st & _lib.ST_TRUNCATED) != 0)[rest].any()), path


def test_contact_world_reports_overflow_instead_of_clamping_silently():
    from lcp_physics_amd import scenes
    from lcp_physics_amd.physics.batched_world import ContactWorld
    from lcp_physics_amd.physics.contacts import GeometryBatch
    w = scenes.make_drop_world(B=32, nbox=4, seed=5)
    geom = GeometryBatch.from_shapes(w["shapes"], 32).to(DEV)
    mk = lambda maxc: ContactWorld(geom, w["p"].to(DEV), w["v"].to(DEV), w["Mdiag"].to(DEV), w["f"].to(DEV), w["rest"].to(DEV),
                                   w["fric"].to(DEV), Je=w["Je"].to(DEV), maxc=maxc)
    ok = mk(16)
    ok.run(60, graph=False)                                           # a settled 4-box stack has 8 contacts: fits
    assert ok.truncated_scenes().numel() == 0
    small = mk(4)
    with pytest.raises(RuntimeError, match="exceeded maxc"):
        small.run(60, graph=False)
    assert small.truncated_scenes().numel() > 0
