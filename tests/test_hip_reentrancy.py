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


# ------------------------------------------------------------------ kernel family of a backward = its forward's, on any thread
def _fwd_bwd_on_threads(path, bwd_in_thread):
    """Fused step under `path` on this thread; the two backwards (dense gradients, physical gradients) either here or on a
    fresh thread whose path default was never set (what torch's autograd worker thread is)."""
    from lcp_physics_amd import _lib
    from lcp_physics_amd.lcp import lcp_backward
    from lcp_physics_amd.physics import assemble_contacts, fused_step
    from lcp_physics_amd.physics.batched_world import fused_step_backward, solution_of_step
    sc = _scenes(21, B=256)
    lcp = assemble_contacts(sc)
    cot = torch.randn(sc.B, 3 * sc.nb, generator=torch.Generator().manual_seed(5)).to(DEV)
    _lib.set_path(path)
    try:
        out = fused_step(sc)
    finally:
        _lib.set_path("auto")
    res = {}

    def bwd():
        with torch.cuda.device(0):
            res["dense"] = lcp_backward(solution_of_step(sc, out, lcp[2], lcp[4]), cot)
            res["phys"] = fused_step_backward(sc, out, (-cot).reshape(sc.B, sc.nb, 3).contiguous())
            torch.cuda.synchronize()
    if bwd_in_thread:
        t = threading.Thread(target=bwd); t.start(); t.join()
    else:
        bwd()
    return out, res


@pytest.mark.parametrize("path", ["auto", "big"])
def test_backward_on_another_thread_picks_the_forwards_kernel_family(path):
    """the kernel family (and with it the workspace layout) used to be re-derived from thread-local state at backward
    time.  Now it is a function of the `compute` word recorded at forward time: the backward run on another host thread is
    bitwise the one run on the forward's thread - for the body-space layout (no W in the workspace) and the contact-space one."""
    out_a, a = _fwd_bwd_on_threads(path, False)
    out_b, b = _fwd_bwd_on_threads(path, True)
    assert torch.equal(out_a["v_new"], out_b["v_new"])
    for ga, gb in zip(a["dense"], b["dense"]):
        assert torch.equal(ga, gb) and bool(torch.isfinite(ga).all())
    for k in a["phys"]:
        assert torch.equal(a["phys"][k], b["phys"][k]) and bool(torch.isfinite(a["phys"][k]).all()), k


def test_backward_planned_for_another_layout_returns_nan_not_garbage():
    """The forward leaves a layout tag in the workspace trailer; a backward whose `compute` word plans another kernel family /
    layout finds the mismatch on the device and returns NaN gradients (the launch itself cannot fail: nothing synchronises)."""
    from lcp_physics_amd import _lib
    from lcp_physics_amd.lcp import lcp_backward
    from lcp_physics_amd.physics import assemble_contacts, fused_step
    from lcp_physics_amd.physics.batched_world import fused_step_backward, solution_of_step
    sc = _scenes(22, B=64)
    lcp = assemble_contacts(sc)
    cot = torch.ones(sc.B, 3 * sc.nb, device=DEV)
    out = fused_step(sc)                                        # body-space forward: no W in the workspace
    good = fused_step_backward(sc, out, cot.reshape(sc.B, sc.nb, 3))
    wrong = dict(out)
    wrong["compute"] = out["compute"] | _lib.PATH_CONTACT_SPACE  # a backward that would re-factor a W that is not there
    bad = fused_step_backward(sc, wrong, cot.reshape(sc.B, sc.nb, 3))
    bad_dense = lcp_backward(solution_of_step(sc, wrong, lcp[2], lcp[4]), cot)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(good["v"]).all())
    assert bool(torch.isnan(bad["v"]).all()) and bool(torch.isnan(bad_dense[1]).all())
    # and the dense forward's workspace is not a contact-list forward's: LCP_HINT_ALL_CONTACT on it is refused on the device too
    from lcp_physics_amd.lcp import lcp_solve
    sol = lcp_solve(*lcp)
    sol.all_contact = True
    nanp = lcp_backward(sol, cot)[1]
    torch.cuda.synchronize()
    assert bool(torch.isnan(nanp).all())


def test_differentiable_step_on_the_generic_kernels_has_the_gradients_of_the_default_path():
    """a recorded step must not fail with LCP_E_TOOLARGE in the middle of loss.backward(): SolveDynamicsFunction decides
    when the step is RECORDED (`lcp_step_has_backward`).  Since round 6 the generic kernels keep their iterate and have a fused
    backward of their own (`lcp_step_bwd_kernel`): forced onto them, the same scenes give the gradients of the default path."""
    from lcp_physics_amd import _lib
    from lcp_physics_amd.physics.batched_world import SolveDynamicsFunction
    sc = _scenes(23, B=8)
    Md = sc.Mdiag.clone().requires_grad_(True)
    count = torch.full((sc.B,), sc.nc, dtype=torch.int32, device=DEV)
    args = (Md, sc.v, sc.f, sc.rest, sc.fric, sc.c_n, sc.c_p1, sc.c_p2, sc.c_i1, sc.c_i2, count, sc.Je, sc.dt)
    opts = {}
    v_new = SolveDynamicsFunction.apply(*args, opts)
    v_new.sum().backward()                                       # the default path has one
    assert Md.grad is not None and bool(torch.isfinite(Md.grad).all()) and "dense_boundary" not in opts["last"]
    fused, Md.grad = Md.grad.clone(), None
    _lib.set_path("generic")
    try:
        opts = {}
        v_dense = SolveDynamicsFunction.apply(*args, opts)
        assert "dense_boundary" not in opts["last"]
        v_dense.sum().backward()
        with torch.no_grad():
            SolveDynamicsFunction.apply(*args, {})               # forward only: the fused generic step
    finally:
        _lib.set_path("auto")
    assert float((v_dense.detach() - v_new.detach()).abs().max() / v_new.detach().abs().max()) < 1e-5
    scale = fused.abs().reshape(sc.B, -1).max(dim=1)[0].clamp_min(1e-30)
    assert float(((Md.grad - fused).abs().reshape(sc.B, -1).max(dim=1)[0] / scale).median()) < 1e-4
