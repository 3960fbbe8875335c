"""The HIP path at the BASELINE.json configurations, each at its single-GPU share:

  config 2  bump-on-tail, N=1e6, Ng=256,  64 envs, fp64, no control
  config 3  two-stream,   N=1e6, Ng=512, 128 envs, fp32, a new random E_in action every step
  config 4  bump-on-tail, N=4e6, Ng=1024, 64 envs (512 / 8 GPUs), fp64
  config 5  bump-on-tail, N=1e7, Ng=256, 128 envs (1024 / 8 GPUs), fp32 push / fp64 Poisson

fp64 configurations: the first and the last environment against the NumPy oracle after one step
(src/env/pic.py:131-146); environment 0 against the oracle again after all 20 steps (these sizes run with coarser deposit
weights than the small golden trajectories: 2^-42 at N=1e6, 2^-40 at N=4e6 -- the K-step bound is measured x 100);
environment 17 stepped alone in a handle of its own against its copy inside the batch, bit for bit; then size-independent
invariants on every environment.
fp32 configurations: against the fp64 HIP run of the same inputs and actions on every environment and
against the oracle for one environment, with bounds from the measured error model
(profiles/fp32_error_model.md): every bound is a measured value times a stated margin.
No reference fixture covers float32, so for the fp32 modes parity is pinned only through the fp64 path.

Ensembles are drawn by the device sampler (pic_reset_sampled) so that no test moves gigabytes over PCIe;
what is compared with the oracle is downloaded per environment through the zero-copy torch views.
"""
import numpy as np
import pytest

from conftest import circ_err, record_measure, rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def oc():
    import ocplasma_amd
    return ocplasma_amd


@pytest.fixture(scope="module")
def po():
    from oracle import pic_oracle
    return pic_oracle


def env_state(env, e):
    """(x, v) of environment e as float64 host arrays, through the device views."""
    env.sync()                      # before the views: the float copy of fixed-point positions is taken at call time
    t = env.torch_views()
    return t["x"][e].double().cpu().numpy(), t["v"][e].double().cpu().numpy()


def oracle_after(po, x0, v0, Ng, L, dt, exts):
    """Oracle state after len(exts) steps from (x0, v0); exts[k] is the (Ng,) external field of step k or None."""
    ref = po.OraclePIC(x0, v0, Ng, L=L, dt=dt, perturb=False, faithful=False)
    for ext in exts:
        ref.update_state(None if ext is None else np.asarray(ext, dtype=float).reshape(-1, 1))
    return ref


def check_invariants(env, N, Ng, L, n0, ke0, pe0, tag, drift_tol, charge_tol=1e-9):
    """Size-independent properties of every environment (DESIGN.md 'parity at full size')."""
    import torch
    dx = L / Ng
    env.sync()
    t = env.torch_views()
    x, v = t["x"], t["v"]
    assert bool((x >= 0).all()) and bool((x < L).all()) and env.bad_count() == 0
    n, Em, phi = env.fields()
    ke, pe, per = env.energies()
    charge = float(np.max(np.abs(n.sum(axis=1) * dx - n0 * L)))
    lap_phi = (np.roll(phi, -1, 1) - 2 * phi + np.roll(phi, 1, 1)) / dx ** 2
    poisson = float(np.max(np.abs(lap_phi - (n - n0))))
    grad_phi = (np.roll(phi, -1, 1) - np.roll(phi, 1, 1)) / (2 * dx)
    efield = float(np.max(np.abs(Em + grad_phi)))
    drift = float(np.max(np.abs((ke + pe) / (ke0 + pe0) - 1)))
    ke_dev = 0.5 * (v.double() ** 2).sum(dim=1).cpu().numpy()
    record_measure(f"{tag}.charge_err", charge)
    record_measure(f"{tag}.poisson_residual", poisson)
    record_measure(f"{tag}.energy_drift_20_steps", drift)
    assert charge < charge_tol, charge
    assert float(np.max(np.abs(Em.mean(axis=1)))) < 1e-11 and float(np.max(np.abs(phi.mean(axis=1)))) < 1e-11
    assert poisson < 1e-8 and efield < 1e-10
    assert np.allclose(per * N / L, pe, rtol=1e-14)
    assert np.allclose(ke_dev, ke, rtol=1e-10 if x.dtype == torch.float64 else 1e-6)
    assert drift < drift_tol, drift


def fp64_config(oc, po, tag, kind, E_, N, Ng, seed):
    L, n0 = 50.0, 1.0
    env = oc.BatchedPIC(E_, N, Ng, L=L, dt=0.1)
    assert abs(env.dt - min(0.1, 2 / np.sqrt(N / L))) < 1e-18            # CFL clamp (pic.py:71-73)
    # a state of this size lives in HBM: pic_create looked for x and v in two different regions of it (include/picstep.h)
    tried, kept, slowest, secs = env._h.placement_info()
    assert 1 <= tried <= 64 and (tried == 1 or kept >= slowest > 0.0) and secs < 1.0
    record_measure(f"{tag}.placement.candidates", tried)
    record_measure(f"{tag}.placement.kept_GBs", kept)
    record_measure(f"{tag}.placement.slowest_GBs", slowest)
    record_measure(f"{tag}.placement.seconds", secs)
    env.reset_sampled(kind, seed=seed)
    ke0, pe0, _ = env.energies()
    picks = (0, E_ - 1)
    lone = 17 % E_
    start = {e: env_state(env, e) for e in set(picks) | {lone}}
    env.step()
    n, Em, phi = env.fields()
    ke, pe, per = env.energies()
    worst = {}
    for e in picks:
        ref = oracle_after(po, *start[e], Ng, L, 0.1, [None])
        assert ref.dt == env.dt
        x, v = env_state(env, e)
        errs = {"x": circ_err(x, ref.x, L) / L, "v": rel_err(v, ref.v), "n": rel_err(n[e], ref.n),
                "E_mesh": rel_err(Em[e], ref.E_mesh), "H": abs((ke[e] + pe[e]) / ref.get_energy() - 1)}
        for k, val in errs.items():
            worst[k] = max(worst.get(k, 0.0), val)
    for k, val in worst.items():
        record_measure(f"{tag}.one_step.{k}", val)
    # north star: 1e-6 relative; one step of the fp64 path sits many orders below it
    assert worst["x"] < 1e-12 and worst["v"] < 1e-12 and worst["n"] < 1e-12 and worst["E_mesh"] < 1e-9
    assert worst["H"] < 1e-12
    env.step(None, nsteps=19)
    check_invariants(env, N, Ng, L, n0, ke0, pe0, tag, drift_tol=1e-6)
    n, Em, phi = env.fields()
    ke, pe, per = env.energies()
    # K-step parity at full size: environment 0 after the 20 steps against 20 oracle steps
    ref = oracle_after(po, *start[0], Ng, L, 0.1, [None] * 20)
    x, v = env_state(env, 0)
    errs = {"x": circ_err(x, ref.x, L) / L, "v": rel_err(v, ref.v), "n": rel_err(n[0], ref.n),
            "E_mesh": rel_err(Em[0], ref.E_mesh), "H": abs((ke[0] + pe[0]) / ref.get_energy() - 1)}
    for k, val in errs.items():
        record_measure(f"{tag}.20_steps.{k}", val)
    bound = K20_BOUNDS[tag]
    assert all(errs[k] < bound[k] for k in errs), (errs, bound)
    # one environment of the batch stepped alone (another launch geometry, its own accumulator sub-rows): the same bits
    solo = oc.BatchedPIC(1, N, Ng, L=L, dt=0.1)
    solo.reset(start[lone][0][None], start[lone][1][None])
    solo.step(None, nsteps=20)
    xs, vs = env_state(solo, 0)
    xb, vb = env_state(env, lone)
    ns, Es, ps = solo.fields()
    assert np.array_equal(xs, xb) and np.array_equal(vs, vb)
    assert np.array_equal(ns[0], n[lone]) and np.array_equal(Es[0], Em[lone]) and np.array_equal(ps[0], phi[lone])
    kes, pes, _ = solo.energies()
    assert pes[0] == pe[lone] and abs(kes[0] / ke[lone] - 1) < 1e-14
    solo.close()
    env.close()


# 20 steps of environment 0 against the oracle: measured on MI355X (profiles/r3_measured_errors.json: config 2 x 5.7e-16 of L,
# v 8.1e-16, n 3.6e-15, E_mesh 5.5e-14, H 2.2e-16; config 4 x 5.7e-16, v 1.3e-15, n 2.0e-14, E_mesh 7.8e-13, H 2.2e-16) x 100.
# The deposit weights of these sizes are rounded to 2^-42 / 2^-40 (fg); over 20 steps that stays at the level of float64 itself.
K20_BOUNDS = {"config2": {"x": 1e-13, "v": 1e-13, "n": 4e-13, "E_mesh": 6e-12, "H": 3e-14},
              "config4": {"x": 1e-13, "v": 2e-13, "n": 2e-12, "E_mesh": 8e-11, "H": 3e-14}}


def test_config2_bump_on_tail_1e6_256_64envs_fp64(oc, po):
    fp64_config(oc, po, "config2", "bump-on-tail", 64, 1_000_000, 256, seed=2)


def test_config4_share_bump_on_tail_4e6_1024_64envs_fp64(oc, po):
    fp64_config(oc, po, "config4", "bump-on-tail", 64, 4_000_000, 1024, seed=4)


def fp32_config(oc, po, tag, kind, E_, N, Ng, seed, with_actions, bounds, position_dtype=None):
    """fp32 particles against (i) the fp64 HIP path on every environment, same float32-representable start
    and the same per-step actions, (ii) the oracle for environment 0 after 1, 10 and 20 steps.  No reference fixture can
    cover these modes (the reference is float64 only): beyond this, parity of the float32 formats is pinned through the
    float64 path."""
    import torch
    L, n0, M = 50.0, 1.0, 3
    kw = {} if position_dtype is None else {"position_dtype": position_dtype}
    lo = oc.BatchedPIC(E_, N, Ng, L=L, dt=0.1, dtype="float32", **kw)
    hi = oc.BatchedPIC(E_, N, Ng, L=L, dt=0.1)
    hi.reset_sampled(kind, seed=seed)
    t = hi.torch_views()
    hi.sync()
    x32 = t["x"].float().contiguous()
    v32 = t["v"].float().contiguous()
    x32 = torch.where(x32 >= L, torch.zeros_like(x32), x32)      # float32 rounding of a value just below L
    x64, v64 = x32.double().contiguous(), v32.double().contiguous()
    torch.cuda.synchronize()
    lo.reset_device(x32.data_ptr(), v32.data_ptr())
    hi.reset_device(x64.data_ptr(), v64.data_ptr())
    lo.sync(); hi.sync()
    x0, v0 = x64[0].cpu().numpy(), v64[0].cpu().numpy()
    del x32, v32, x64, v64, t
    torch.cuda.empty_cache()
    act = oc.E_field(L, Ng, M)
    if with_actions:
        lo.set_actuator(act)
        hi.set_actuator(act)
    ke0, pe0, _ = lo.energies()
    rng = np.random.default_rng(seed)
    # the oracle follows environment 0 through all 20 steps (round 4; rounds 2-3 compared after the first step only)
    ref = po.OraclePIC(x0, v0, Ng, L=L, dt=0.1, perturb=False, faithful=False)
    checkpoints = {1, 10, 20}
    for k in range(1, 21):
        if with_actions:
            a = rng.uniform(-1.25, 1.25, (E_, 2 * M))             # a new action every step
            lo.step_actions(a)
            hi.step_actions(a)
            ref.update_state(act.compute_E_batched(a)[0].reshape(-1, 1))
        else:
            lo.step()
            hi.step()
            ref.update_state(None)
        if k not in checkpoints:
            continue
        lo.sync(); hi.sync()
        tl, th = lo.torch_views(), hi.torch_views()
        dx_ = (tl["x"].double() - th["x"]).abs()
        ex = float(torch.minimum(dx_, L - dx_).max()) / L
        ev = float((tl["v"].double() - th["v"]).abs().max() / th["v"].abs().max())
        (nl, El, _), (nh, Eh, _) = lo.fields(), hi.fields()
        eE = max(rel_err(El[e], Eh[e]) for e in range(E_))
        en = max(rel_err(nl[e], nh[e]) for e in range(E_))
        (kl, pl, _), (kh, ph, _) = lo.energies(), hi.energies()
        eke = float(np.max(np.abs(kl / kh - 1)))
        epe = float(np.max(np.abs(pl / ph - 1)))
        for name, val in (("x", ex), ("v", ev), ("n", en), ("E_mesh", eE), ("KE", eke), ("PE", epe)):
            record_measure(f"{tag}.vs_fp64.step{k}.{name}", val)
            assert val < bounds[k][name], (k, name, val)
        # ... and environment 0 against the oracle (float64 NumPy restatement of the reference) at every checkpoint, at the
        # bounds held against the float64 HIP run (which follows the oracle to 1e-13: the two errors are the same error)
        xo, vo = env_state(lo, 0)
        eo = {"x": circ_err(xo, ref.x, L) / L, "v": rel_err(vo, ref.v), "E_mesh": rel_err(El[0], ref.E_mesh),
              "n": rel_err(nl[0], ref.n), "PE": abs(pl[0] / ref.get_electric_energy() - 1)}
        for name, val in eo.items():
            record_measure(f"{tag}.vs_oracle.step{k}.{name}", val)
            assert val < bounds[k][name], (k, name, val)
    dxm = L / Ng
    n, _, _ = lo.fields()
    charge = float(np.max(np.abs(n.sum(axis=1) * dxm - n0 * L)))
    record_measure(f"{tag}.charge_err", charge)
    assert charge < 1e-10                       # integer deposit sums: the total charge is exact
    ke, pe, _ = lo.energies()
    (kh, ph, _) = hi.energies()
    drift_lo = float(np.max(np.abs((ke + pe) / (ke0 + pe0) - 1)))
    record_measure(f"{tag}.energy_change_20_steps_fp32", drift_lo)
    record_measure(f"{tag}.energy_gap_to_fp64_20_steps", float(np.max(np.abs((ke + pe) / (kh + ph) - 1))))
    assert float(np.max(np.abs((ke + pe) / (kh + ph) - 1))) < bounds["H_gap"]
    lo.sync()
    tl = lo.torch_views()
    assert bool((tl["x"] >= 0).all()) and bool((tl["x"] < L).all()) and lo.bad_count() == 0
    lo.close()
    hi.close()


# Bounds = 3-4 x the worst value measured on MI355X over configs 3 and 5, all 128 environments (round-2 collection,
# profiles/r2_measured_errors.json; the CPU emulation profiles/fp32_error_model.md predicts the same orders).  Errors are
# max-norm relative to the float64 HIP run: x on the circle relative to L, the others to the max of the reference.
#   float32 positions: the 3.8e-6 position ulp near x = L dominates (E_mesh 6.5e-6 after one step, 8.4e-5 after 20);
#   fixed-point positions (1.2e-8 resolution): what is left is the float32 velocity (v 1.3e-7 per step) -- E_mesh 2.9e-8
#   after one step, 7.2e-7 after 20.
FP32_BOUNDS = {
    1: {"x": 5e-7, "v": 5e-7, "n": 1e-5, "E_mesh": 2e-5, "KE": 5e-9, "PE": 4e-6},
    10: {"x": 5e-6, "v": 2e-6, "n": 4e-5, "E_mesh": 1.5e-4, "KE": 1e-8, "PE": 2e-5},
    20: {"x": 1e-5, "v": 4e-6, "n": 5e-5, "E_mesh": 3e-4, "KE": 2e-8, "PE": 3e-5},
    "H_gap": 2e-8,
}
FIXED32_BOUNDS = {
    1: {"x": 4e-9, "v": 5e-7, "n": 5e-8, "E_mesh": 1e-7, "KE": 5e-9, "PE": 3e-8},
    10: {"x": 3e-8, "v": 2e-6, "n": 4e-7, "E_mesh": 1e-6, "KE": 1e-8, "PE": 1e-6},
    20: {"x": 8e-8, "v": 4e-6, "n": 7e-7, "E_mesh": 2.5e-6, "KE": 2e-8, "PE": 2e-6},
    "H_gap": 2e-8,
}


def test_config3_two_stream_1e6_512_128envs_fp32_random_actions(oc, po):
    fp32_config(oc, po, "config3", "two-stream", 128, 1_000_000, 512, seed=3, with_actions=True, bounds=FP32_BOUNDS)


def test_config5_share_bump_on_tail_1e7_256_128envs_fp32_push_fp64_poisson(oc, po):
    fp32_config(oc, po, "config5", "bump-on-tail", 128, 10_000_000, 256, seed=5, with_actions=False,
                bounds=FP32_BOUNDS)


def test_config3_fixed_point_positions(oc, po):
    fp32_config(oc, po, "config3_fixed32", "two-stream", 128, 1_000_000, 512, seed=3, with_actions=True,
                bounds=FIXED32_BOUNDS, position_dtype="fixed32")


def test_config5_share_fixed_point_positions(oc, po):
    fp32_config(oc, po, "config5_fixed32", "bump-on-tail", 128, 10_000_000, 256, seed=5, with_actions=False,
                bounds=FIXED32_BOUNDS, position_dtype="fixed32")


def test_small_states_skip_the_placement_comparison(oc):
    """Below 256 MB of particles the state sits in the Infinity Cache or the step is latency-bound: x | v in one block, nothing timed."""
    env = oc.BatchedPIC(3, 5000, 250, L=50.0, dt=0.1)
    assert env._h.placement_info() == (1, 0.0, 0.0, 0.0)
    env.close()
