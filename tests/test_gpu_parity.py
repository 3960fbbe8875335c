"""Parity of the HIP path (through the C ABI) with the reference: golden vectors produced by the
reference itself (tests/golden) and the NumPy oracle on seeded inputs.

Tolerances (north star: <= 1e-6 relative, fp64): single evaluations are checked at 1e-12, K-step
trajectories at bounds that follow the measured growth of rounding differences in this chaotic
system (a mere particle permutation of the reference moves E_mesh by 1e-12 after 500 steps).
Positions are compared on the circle (x = 0 and x = L - eps are neighbours).
"""
import numpy as np
import pytest

from conftest import circ_err, load_golden, record_measure, rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def oc():
    import ocplasma_amd
    return ocplasma_amd


@pytest.fixture(scope="module")
def po():
    from oracle import pic_oracle
    return pic_oracle


class FixedDist:
    """init_dist stand-in that hands PIC the particles a golden file recorded (pic.py:64-65)."""

    def __init__(self, x, v):
        self.x_init, self.v_init = np.array(x, dtype=float).ravel(), np.array(v, dtype=float).ravel()
        self.n_samples = self.x_init.size

    def reinit(self):
        pass

    def get_sample(self):
        return self.x_init.copy(), self.v_init.copy()

    def get_init_state(self):
        return np.concatenate([self.x_init.reshape(-1, 1), self.v_init.reshape(-1, 1)], axis=0)


def make_pic(oc, g, interpol="CIC", dtype="float64"):
    return oc.PIC(N=int(g["N"]), N_mesh=int(g["Ng"]), n0=float(g["n0"]), L=float(g["L"]), dt=float(g["dt_in"]),
                  gamma=float(g["gamma"]), A=float(g["A"]), n_mode=int(g["n_mode"]), interpol=interpol,
                  init_dist=FixedDist(g["x0_raw"], g["v0_raw"]), dtype=dtype)


# ---------------------------------------------------------------------------------------------
# single evaluations against the reference's own outputs
# ---------------------------------------------------------------------------------------------
def test_g1_deposit_edge_inputs(oc):
    g = load_golden("g1_deposit")
    L, Ng, n0 = float(g["L"]), int(g["Ng"]), float(g["n0"])
    x = g["x"][:, 0]
    h = oc._abi.Handle(x.size, Ng, 1, L, n0, 0.1)
    n, E, pe = h.eval_field(x[None])
    assert rel_err(n[0], g["n"]) < 1e-13
    # reset() stores the doubly wrapped positions (pic.py:139 + util.py:51) and reports CIC bookkeeping
    h.reset(x[None], np.zeros_like(x)[None])
    xw, _ = h.particles()
    assert np.array_equal(xw[0], g["x_wrapped"][:, 0] % L)      # reference leaves exactly-L values for CIC's own mod
    jl, jr, wl, wr = h.cic(0)
    assert np.array_equal(jl, g["jl"][:, 0]) and np.array_equal(jr, g["jr"][:, 0])
    assert np.array_equal(wl, g["wl"][:, 0]) and np.array_equal(wr, g["wr"][:, 0])
    assert h.bad_count() == 0
    ht = oc._abi.Handle(x.size, Ng, 1, L, n0, 0.1, interpol="TSC")
    nt, _, _ = ht.eval_field(x[None])
    assert rel_err(nt[0], g["tsc_n"]) < 1e-13


def test_g3_compute_E(oc):
    g = load_golden("g3_compute_E")
    L, Ng, n0 = float(g["L"]), int(g["Ng"]), float(g["n0"])
    x = g["x"][:, 0]
    N = x.size
    h = oc._abi.Handle(N, Ng, 1, L, n0, 0.1)
    n, E, pe = h.eval_field(x[None])
    assert rel_err(E[0], g["E_mesh"]) < 1e-12
    assert abs(pe[0] * N / L / float(g["PE"]) - 1) < 1e-12
    n, E, pe = h.eval_field(x[None], g["E_ext"])
    assert rel_err(E[0], g["E_mesh_with_ext"]) < 1e-12
    # gather at the particles + zero-mean phi
    h.reset(x[None], (0.5 * x)[None])
    assert rel_err(h.gather_E()[0], g["E"]) < 1e-12
    _, Em, phi = h.fields()
    ref_phi = g["phi_mesh"][:, 0]
    assert rel_err(phi[0], ref_phi - ref_phi.mean()) < 1e-10
    ke, pe2, _ = h.energies()
    assert abs((ke[0] + pe2[0]) / float(g["H"]) - 1) < 1e-13
    ht = oc._abi.Handle(N, Ng, 1, L, n0, 0.1, interpol="TSC")
    _, Et, _ = ht.eval_field(x[None], g["E_ext"])
    assert rel_err(Et[0], g["tsc_E_mesh_with_ext"]) < 1e-12


def test_g2_mesh_sizes_field_solve(oc, po):
    """Field from a density for Ng in {128, 250, 256, 1024} (reference n -> E in g2_solve)."""
    g = load_golden("g2_solve")
    L, n0 = float(g["L"]), float(g["n0"])
    rng = np.random.default_rng(5)
    for Ng in (128, 250, 256, 1024):
        x = rng.uniform(0, L, 20000)
        h = oc._abi.Handle(x.size, Ng, 1, L, n0, 0.1)
        n, E, _ = h.eval_field(x[None])
        u = x.reshape(-1, 1).copy()
        _, Eo = po.field_at_particles(u, L / Ng, Ng, n0, L, x.size)
        assert rel_err(E[0], Eo) < 5e-12, Ng
        # the reference's own density through the device solver (pic_solve_poisson) against the field and the potential the
        # reference derived from it, for both of its gammas (the gauge of its phi is round-off: compared mean-removed)
        phi_d, E_d = h.solve_poisson(g[f"n_{Ng}"] - n0)
        for gam, tol in (("5.0", 5e-12), ("0.3", 5e-11)):
            assert rel_err(E_d, g[f"E_{Ng}_g{gam}"]) < tol, (Ng, gam)
        ref_phi = g[f"phi_{Ng}_g5.0"]
        assert rel_err(phi_d, ref_phi - ref_phi.mean()) < 1e-9, Ng
        h.close()


# ---------------------------------------------------------------------------------------------
# trajectories through the PIC drop-in class
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", ["g5_bump_on_tail_N10000_Ng128", "g5_two_stream_N5000_Ng250"])
def test_g5_trajectory(oc, name):
    g = load_golden(name)
    L = float(g["L"])
    sim = make_pic(oc, g)
    assert sim.dt == float(g["dt"])
    assert np.array_equal(sim.x, g["x_init"]) and np.array_equal(sim.v, g["v_init"])
    assert rel_err(sim.E_mesh, g["E_mesh_init"]) < 1e-12 and rel_err(sim.E, g["E_init"]) < 1e-12
    rew = oc.Reward(sim.init_dist.get_init_state(), int(g["Ng"]), L, -25.0, 25.0, 1.0, 1.0, 1.0)
    H, PE, KE, PEr, R = [sim.get_energy()], [sim.get_electric_energy()], [sim.get_kinetic_energy()], [], []
    # (x and v, n, E_mesh) after K steps: 100 x the error measured on MI355X for the worse of the two cases (two-stream:
    # 2.8e-16 / 4.3e-15 / 1.0e-13 at K=1 ... 8.2e-10 / 4.0e-9 / 3.8e-10 at K=500, gpurun_out/measured_r2.json; a mere
    # particle permutation of the reference moves its own E_mesh by 1e-12 after 500 steps).  The step is
    # bitwise reproducible, so these errors do not vary from run to run.  North star: 1e-6.
    tol = {1: (1e-13, 5e-13, 2e-11), 10: (1e-12, 1e-11, 1e-11), 100: (1e-11, 2e-10, 5e-11), 500: (1e-7, 5e-7, 5e-8)}
    for k in range(1, 501):
        if k <= 100:
            st = sim.get_state()
            R.append(rew.compute_reward(st, g["actions"][k - 1]))
            PEr.append(sim.get_reward_electric_energy())
        sim.update_state(None)
        H.append(sim.get_energy()); PE.append(sim.get_electric_energy()); KE.append(sim.get_kinetic_energy())
        if k in tol:
            t_xv, t_n, t_E = tol[k]
            errs = {"x": circ_err(sim.x, g[f"x_{k}"], L) / L, "v": rel_err(sim.v, g[f"v_{k}"]),
                    "E_mesh": rel_err(sim.E_mesh, g[f"E_mesh_{k}"]), "n": rel_err(sim.n, g[f"n_{k}"])}
            for q, val in errs.items():
                record_measure(f"g5.{name}.step{k}.{q}", val)
            assert errs["x"] < t_xv and errs["v"] < t_xv and errs["n"] < t_n and errs["E_mesh"] < t_E, (k, errs)
    for q, ours, ref in (("H", H, g["H"]), ("KE", KE, g["KE"]), ("PE100", PE[:101], g["PE"][:101])):
        record_measure(f"g5.{name}.trace.{q}", rel_err(ours, ref))
    assert rel_err(H, g["H"]) < 1e-10                    # measured 5.3e-13 over 500 steps
    assert rel_err(KE, g["KE"]) < 1e-10                  # 6.8e-13
    assert rel_err(PE[:101], g["PE"][:101]) < 1e-10      # 2.8e-13
    assert rel_err(PEr, g["PE_reward"][:100]) < 1e-10
    assert rel_err(R, g["reward"][:100]) < 1e-10
    sim.close()


@pytest.mark.parametrize("name,interpol", [("g4_bump_on_tail_ext_N4000_Ng256", "CIC"),
                                           ("g4_two_stream_ext_N3000_Ng200", "CIC"),
                                           ("g4_tsc_bump_on_tail_ext_N3000_Ng128", "TSC")])
def test_g4_external_field_steps(oc, name, interpol):
    g = load_golden(name)
    L, Ng = float(g["L"]), int(g["Ng"])
    mm = g["actions"].shape[1] // 2
    sim = make_pic(oc, g, interpol)
    act = oc.E_field(L, Ng, mm)
    rew = oc.Reward(sim.init_dist.get_init_state(), Ng, L, -25.0, 25.0, 1.0, 1.0, 1.0)
    R, H = [], [sim.get_energy()]
    for k in range(1, 21):
        a = g["actions"][k - 1]
        act.update_E(a[:mm], a[mm:])
        R.append(rew.reward_from_energy(sim.get_reward_electric_energy(), a))
        sim.update_state(E_external=act.compute_E())
        H.append(sim.get_energy())
        if k == 1:
            assert circ_err(sim.x, g["x_1"], L) / L < 1e-13 and rel_err(sim.v, g["v_1"]) < 1e-13
            assert rel_err(sim.n, g["n_1"]) < 1e-12 and rel_err(sim.E_mesh, g["E_mesh_1"]) < 1e-11
            assert rel_err(sim.E, g["E_1"]) < 1e-11
            p, pr = sim.phi_mesh, g["phi_mesh_1"]
            assert rel_err(p, pr - pr.mean()) < 1e-9
            assert np.array_equal(sim.indx_l, g["indx_l_1"]) and np.array_equal(sim.indx_r, g["indx_r_1"])
            assert rel_err(sim.weight_l, g["weight_l_1"]) < 1e-9 and rel_err(sim.weight_r, g["weight_r_1"]) < 1e-9
            if interpol == "CIC":
                assert sim.indx_m is None and sim.weight_m is None
            else:                                               # pic.py:98-99, 109-110
                assert np.array_equal(sim.indx_m, (sim.indx_l + 1) % Ng)
                assert np.max(np.abs(sim.weight_l + sim.weight_m + sim.weight_r - 1)) < 1e-14
    assert circ_err(sim.x, g["x_20"], L) / L < 1e-10 and rel_err(sim.v, g["v_20"]) < 1e-10
    assert rel_err(sim.E_mesh, g["E_mesh_20"]) < 1e-9
    assert rel_err(H, g["H"]) < 1e-12
    if interpol == "CIC":
        assert rel_err(R, g["reward"]) < 1e-10
    sim.close()


def test_reward_and_spectrum_on_host_states(oc):
    """Reward.compute_reward / compute_E_k_spectrum fed with host states, as the trainers call them."""
    g = load_golden("g5_two_stream_N5000_Ng250")
    st = np.concatenate([g["x_10"], g["v_10"]], 0)
    rew = oc.Reward(np.concatenate([g["x0_raw"], g["v0_raw"]]).reshape(-1, 1), 250, 50.0, -25.0, 25.0, 1.0, 1.0, 1.0)
    assert abs(rew.compute_electric_energy(st) / g["PE_reward"][10] - 1) < 1e-9
    assert abs(rew.compute_reward(st, g["actions"][10]) - g["reward"][10]) < 1e-10
    s = load_golden("g9_spectrum")
    snap = np.concatenate([st, np.concatenate([g["x_100"], g["v_100"]], 0)], 1)
    ks, Ek = oc.compute_E_k_spectrum(1.0, 50.0, 50.0 / 250, 250, snap, False)
    assert np.allclose(ks[:8], s["ks"], rtol=1e-14) and rel_err(Ek[:8], s["Ek"]) < 1e-10


def test_reference_style_loop_and_gym_aliases(oc):
    """run_wo_oc.py:108-125 shaped loop + reinit semantics (fields read None until the next step)."""
    np.random.seed(3)
    dist = oc.BumpOnTail(a=0.2, v0=3.0, sigma=1.0, n_samples=6000, L=50.0)
    sim = oc.PIC(N=6000, N_mesh=128, n0=1.0, L=50.0, dt=0.1, tmin=0.0, tmax=1.0, gamma=5.0, A=0.1, n_mode=2,
                 interpol="CIC", init_dist=dist)
    E0 = sim.get_energy()
    pos = []
    for _ in range(10):
        sim.update_state(None)
        pos.append(sim.x.copy())
        assert sim.get_state().shape == (12000, 1)
    assert abs(sim.get_energy() / E0 - 1) < 1e-3
    assert pos[0].shape == (6000, 1) and (pos[-1] >= 0).all() and (pos[-1] < 50.0).all()
    sim.reinit()
    assert sim.E is None and sim.E_mesh is None and sim.phi_mesh is None
    obs, r, done, info = sim.step(None)
    assert obs.shape == (12000, 1) and 0.0 <= r <= 1.0 and not done and info["KE"] > 0
    assert sim.E_mesh.shape == (128, 1)
    snap, E, PE = sim.simulate(None)
    assert snap.shape == (12000, 11) and E.shape == (11,) and PE.shape == (11,)
    assert sim.reset().shape == (12000, 1)
    sim.close()


# ---------------------------------------------------------------------------------------------
# against the oracle on seeded synthetic ensembles
# ---------------------------------------------------------------------------------------------
def test_batched_envs_match_oracle_and_single_env(oc, po):
    E_, N, Ng, L = 5, 30011, 256, 50.0          # odd N: exercises the scalar tail and the padded stride
    xs, vs = zip(*[po.synthetic_bump_on_tail(N, L, seed=100 + e) for e in range(E_)])
    x0, v0 = np.stack(xs), np.stack(vs)
    act = oc.E_field(L, Ng, 3)
    rng = np.random.default_rng(1)
    env = oc.BatchedPIC(E_, N, Ng, L=L, dt=0.1)
    env.reset(x0, v0)
    refs = [po.OraclePIC(x0[e], v0[e], Ng, L=L, dt=0.1, perturb=False, faithful=False) for e in range(E_)]
    for k in range(5):
        ext = act.compute_E_batched(rng.uniform(-1.25, 1.25, (E_, 6)))
        env.step(ext)
        for e in range(E_):
            refs[e].update_state(ext[e].reshape(-1, 1))
    x, v = env.particles()
    n, Em, phi = env.fields()
    ke, pe, per = env.energies()
    Ep = env.gather_E()
    for e in range(E_):
        assert circ_err(x[e], refs[e].x, L) / L < 1e-12 and rel_err(v[e], refs[e].v) < 1e-12
        assert rel_err(n[e], refs[e].n) < 1e-12 and rel_err(Em[e], refs[e].E_mesh) < 1e-10
        assert rel_err(Ep[e], refs[e].E) < 1e-10
        assert abs(ke[e] / refs[e].kinetic_energy() - 1) < 1e-13
        assert abs(pe[e] / refs[e].get_electric_energy() - 1) < 1e-9
    assert env.bad_count() == 0
    # environment 3 stepped alone gives the same answer as inside the batch (up to deposit order)
    solo = oc.BatchedPIC(1, N, Ng, L=L, dt=0.1)
    solo.reset(x0[3:4], v0[3:4])
    rng = np.random.default_rng(1)
    for k in range(5):
        solo.step(act.compute_E_batched(rng.uniform(-1.25, 1.25, (E_, 6)))[3:4])
    xs_, vs_ = solo.particles()
    assert circ_err(xs_[0], x[3], L) / L < 1e-13 and rel_err(vs_[0], v[3]) < 1e-13


def test_nsteps_in_one_call_equals_repeated_calls(oc, po):
    N, Ng, L = 20000, 128, 50.0
    x0, v0 = po.synthetic_two_stream(N, L, seed=9)
    a = oc.BatchedPIC(2, N, Ng, L=L, dt=0.05)
    b = oc.BatchedPIC(2, N, Ng, L=L, dt=0.05)
    for env in (a, b):
        env.reset(np.stack([x0, x0[::-1]]), np.stack([v0, v0[::-1]]))
    a.step(None, nsteps=7)
    for _ in range(7):
        b.step(None)
    (xa, va), (xb, vb) = a.particles(), b.particles()
    assert circ_err(xa, xb, L) / L < 1e-13 and rel_err(va, vb) < 1e-13
    # permutation invariance: env 1 holds env 0's particles in reverse order
    assert circ_err(xa[0], xa[1][::-1], L) / L < 1e-12 and rel_err(a.fields()[1][0], a.fields()[1][1]) < 1e-11


@pytest.mark.parametrize("dtype,pos,interpol,accum", [("float64", None, "CIC", None), ("float64", None, "TSC", None),
                                                     ("float32", None, "CIC", None),
                                                     ("float32", "fixed32", "CIC", None), ("float32", None, "TSC", None),
                                                     ("float32", "fixed32", "TSC", None)])
@pytest.mark.parametrize("N", [20000, 9_000_000])
def test_inner_steps_of_a_call_make_every_refresh_from_the_same_sums(oc, po, N, dtype, pos, interpol, accum):
    """Inside a multi-step call of the streaming schedule a step of an HBM-resident state (256 MB of particles and more) ends with
    sweep D2 (no deposit of its final positions) and the next step's sweep B2 makes that deposit from the positions it reads; sweep
    C carries the post-step solve (round 4).  Every step's refresh must still be made, from the same integer sums: the energies of
    EVERY step, the fields and the particles of a K-step call equal those of K one-step calls (which end with the full sweep D)
    bit for bit, in every particle format and shape -- and a second call right behind continues the same way.  (Smaller states
    keep the full sweep D in their inner steps, with the solve riding on the next sweep C: N = 20000 is that path, 4 x 9e6
    particles the D2 / B2 one.)"""
    import torch
    big = N > 1_000_000
    E_, Ng, L, K = (4 if big else 3), 96, 50.0, (4 if big else 6)
    ext = 0.04 * np.random.default_rng(2).normal(size=(E_, Ng))
    kw = dict(L=L, dt=0.1, dtype=dtype, position_dtype=pos, interpol=interpol, accum_dtype=accum, blocks_per_env=0 if big else 3)
    a, b = oc.BatchedPIC(E_, N, Ng, **kw), oc.BatchedPIC(E_, N, Ng, **kw)
    assert a._h.schedule() == "streaming"
    if big:
        assert 2 * E_ * N * a.dtype.itemsize >= 256 << 20
        for env in (a, b):
            env.reset_sampled("two-stream", seed=70)                       # the device sampler: the same particles in both handles
    else:
        xs, vs = zip(*[po.synthetic_two_stream(N, L, seed=70 + e) for e in range(E_)])
        x0, v0 = np.stack(xs).astype(dtype), np.stack(vs).astype(dtype)
        x0[x0 >= L] = 0.0
        for env in (a, b):
            env.reset(x0, v0)

    def same_state():
        if not big:
            return all(np.array_equal(p, q) for p, q in zip(a.particles() + a.fields(), b.particles() + b.fields()))
        a.sync(); b.sync()
        ta, tb = a.torch_views(), b.torch_views()                          # (views: no 300 MB read-back)
        key = "x_fixed" if pos == "fixed32" else "x"
        return bool(torch.equal(ta[key], tb[key])) and bool(torch.equal(ta["v"], tb["v"])) and \
            all(np.array_equal(p, q) for p, q in zip(a.fields(), b.fields()))

    for field in (None, ext):
        ke, pe, per = a.step_history(field, K)
        for s in range(K):
            b.step(field)
            kb, pb, rb = b.energies()
            assert np.array_equal(pe[s], pb) and np.array_equal(per[s], rb), (s, field is None)
            assert np.allclose(ke[s], kb, rtol=1e-14)                      # (a float64 sum over workgroups in a fixed order)
        assert same_state()
    a.step(None, K)                                                        # without history: the same schedule, nothing recorded
    for _ in range(K):
        b.step(None)
    assert same_state()
    assert a.bad_count() == 0
    a.close()
    b.close()


def test_energy_history_equals_stepwise_reads(oc, po):
    """pic_step_history / BatchedPIC.simulate: the E and PE traces of PIC.simulate, one read-back for all steps."""
    E_, N, Ng, L, K = 3, 20000, 128, 50.0, 12
    xs, vs = zip(*[po.synthetic_bump_on_tail(N, L, seed=40 + e) for e in range(E_)])
    ext = 0.05 * np.random.default_rng(3).normal(size=(E_, Ng))
    a = oc.BatchedPIC(E_, N, Ng, L=L, dt=0.1)
    b = oc.BatchedPIC(E_, N, Ng, L=L, dt=0.1)
    for env in (a, b):
        env.reset(np.stack(xs), np.stack(vs))
    H, PE = a.simulate(K, ext)
    assert H.shape == PE.shape == (K + 1, E_)
    ke0, pe0, _ = b.energies()
    # two handles agree bit for bit: every deposit is an integer sum
    assert np.array_equal(H[0], ke0 + pe0) and np.array_equal(PE[0], pe0)
    for s in range(1, K + 1):
        b.step(ext)
        ke, pe, per = b.energies()
        assert np.array_equal(H[s], ke + pe) and np.array_equal(PE[s], pe), s
    xa, va = a.particles()
    xb, vb = b.particles()
    assert np.array_equal(xa, xb) and np.array_equal(va, vb)
    ke, pe, per = a.step_history(None, 2)
    assert ke.shape == (2, E_) and np.array_equal(per[-1], a.energies()[2])     # same handle, same numbers
    # against the oracle's own trace for environment 0
    ref = po.OraclePIC(xs[0], vs[0], Ng, L=L, dt=0.1, perturb=False, faithful=False)
    Href = [ref.get_energy()]
    for _ in range(K):
        ref.update_state(ext[0].reshape(-1, 1))
        Href.append(ref.get_energy())
    assert rel_err(H[:, 0], np.array(Href)) < 1e-12


def test_sweep_A_is_skipped_only_when_its_deposit_is_already_there(oc, po):
    """Sweep D / reset also deposit the next step's q1, so sweep A normally never runs; loading
    particles without a refresh invalidates that deposit and must fall back to sweep A."""
    N, Ng, L = 40000, 256, 50.0
    x0, v0 = po.synthetic_bump_on_tail(N, L, seed=21)
    a = oc._abi.Handle(N, Ng, 1, L, 1.0, 0.05)
    b = oc._abi.Handle(N, Ng, 1, L, 1.0, 0.05)
    a.reset(x0[None], v0[None])
    b.reset(x0[None], v0[None])
    a.profile(True)
    b.profile(True)
    for _ in range(3):
        a.step()
        xb, vb = b.particles()
        b.set_particles(xb, vb)         # same state, but the q1 deposit is now stale -> sweep A path
        b.step()
    (xa, va), (xb, vb) = a.particles(), b.particles()
    assert circ_err(xa, xb, L) / L < 1e-13 and rel_err(va, vb) < 1e-13
    assert rel_err(a.fields()[1], b.fields()[1]) < 1e-11
    pa, pb = a.profile_read(), b.profile_read()
    assert "sweep_A" not in pa and pb["sweep_A"][1] == 3
    ref = po.OraclePIC(x0, v0, Ng, L=L, dt=0.05, perturb=False, faithful=False)
    for _ in range(3):
        ref.update_state(None)
    assert circ_err(xa[0], ref.x, L) / L < 1e-13 and rel_err(va[0], ref.v) < 1e-13


def test_full_size_invariants_and_one_step_parity(oc, po):
    """BASELINE config-2 sized environments (N = 1e6, Ng = 256): one-step parity with the oracle on
    environment 0, then size-independent properties over 20 steps."""
    E_, N, Ng, L, n0 = 4, 1_000_000, 256, 50.0, 1.0
    dx = L / Ng
    xs, vs = zip(*[po.synthetic_bump_on_tail(N, L, seed=1234 + e) for e in range(E_)])
    x0, v0 = np.stack(xs), np.stack(vs)
    env = oc.BatchedPIC(E_, N, Ng, L=L, dt=0.1)
    assert abs(env.dt - 2 / np.sqrt(N / L)) < 1e-18            # CFL clamp (pic.py:71-73)
    env.reset(x0, v0)
    ref = po.OraclePIC(x0[0], v0[0], Ng, L=L, dt=0.1, perturb=False, faithful=False)
    ke0, pe0, _ = env.energies()
    assert abs((ke0[0] + pe0[0]) / ref.get_energy() - 1) < 1e-13
    env.step()
    ref.update_state(None)
    x, v = env.particles()
    n, Em, phi = env.fields()
    assert circ_err(x[0], ref.x, L) / L < 1e-13 and rel_err(v[0], ref.v) < 1e-13
    assert rel_err(n[0], ref.n) < 1e-12 and rel_err(Em[0], ref.E_mesh) < 1e-9
    env.step(None, nsteps=19)
    x, v = env.particles()
    n, Em, phi = env.fields()
    ke, pe, per = env.energies()
    assert (x >= 0).all() and (x < L).all() and env.bad_count() == 0
    assert np.max(np.abs(n.sum(axis=1) * dx - n0 * L)) < 1e-9                 # charge conservation
    assert np.max(np.abs(Em.mean(axis=1))) < 1e-12 and np.max(np.abs(phi.mean(axis=1))) < 1e-12
    lap_phi = (np.roll(phi, -1, 1) - 2 * phi + np.roll(phi, 1, 1)) / dx ** 2
    assert np.max(np.abs(lap_phi - (n - n0))) < 1e-9                          # Poisson residual
    grad_phi = (np.roll(phi, -1, 1) - np.roll(phi, 1, 1)) / (2 * dx)
    assert np.max(np.abs(Em + grad_phi)) < 1e-11
    assert np.allclose(per * N / L, pe, rtol=1e-14)
    assert np.allclose(0.5 * (v * v).sum(axis=1), ke, rtol=1e-12)
    assert np.max(np.abs((ke + pe) / (ke0 + pe0) - 1)) < 1e-6                 # energy drift over 20 steps


def test_float32_modes_track_float64(oc, po):
    """float32 particles with every accumulator and position format against the float64 path from the same
    float32-representable start.  The bounds are 2-3x what was measured on MI355X (gpurun_out/measured_r2.json,
    profiles/fp32_error_model.md: float positions lose ~4e-6 of E_mesh per step to the 3.8e-6 position ulp near
    x = L; fixed-point positions are limited by the float32 velocities instead)."""
    N, Ng, L = 200_000, 512, 50.0
    x0, v0 = po.synthetic_two_stream(N, L, seed=5)
    x0, v0 = x0.astype(np.float32), v0.astype(np.float32)
    x0[x0 >= L] = 0.0
    ref = oc.BatchedPIC(1, N, Ng, L=L, dt=0.1)
    ref.reset(x0[None].astype(np.float64), v0[None].astype(np.float64))
    ref.step(None, 10)
    _, Er, _ = ref.fields()
    xr, vr = ref.particles()
    kr, pr, _ = ref.energies()
    # 3-4 x measured (float: E 1.6e-5, x 7.5e-7, v 3.7e-7, PE 3.2e-7; fixed32: E 4.2e-7, v 3.9e-7, PE 2.2e-7)
    bounds = {"float": dict(E=5e-5, x=2.5e-6, v=1.5e-6, pe=1.5e-6), "fixed32": dict(E=1.5e-6, x=5e-8, v=1.5e-6, pe=1e-6)}
    for pos in ("float", "fixed32"):
        for acc in (None, "fixed", "fix64"):
            env = oc.BatchedPIC(1, N, Ng, L=L, dt=0.1, dtype="float32", accum_dtype=acc, position_dtype=pos)
            env.reset(x0[None], v0[None])
            env.step(None, 10)
            x, v = env.particles()
            assert x.dtype == np.float32 and (x >= 0).all() and (x < L).all() and env.bad_count() == 0
            _, E, _ = env.fields()
            ke, pe, _ = env.energies()
            errs = dict(E=rel_err(E, Er), x=circ_err(x, xr, L) / L, v=rel_err(v, vr), pe=abs(pe[0] / pr[0] - 1))
            for q, val in errs.items():
                record_measure(f"f32_modes.{pos}.{acc}.{q}", val)
            b = bounds[pos]
            assert errs["E"] < b["E"] and errs["v"] < b["v"] and errs["pe"] < b["pe"], (pos, acc, errs)
            if pos == "float":
                assert errs["x"] < b["x"], (pos, acc, errs)
            else:   # returned as float32: the comparison sees the float32 rounding of the output (<= 3.8e-6 / 50 / 2)
                assert errs["x"] < 5e-8 + b["x"], (pos, acc, errs)
            assert abs(ke[0] / kr[0] - 1) < 1e-6
            env.close()


def test_packed_accumulator_matches_fix64_accumulator(oc, po):
    """The packed word deposits (count, sum of w_r) per cell in one integer LDS atomic; the density it yields is
    the one of separately accumulated weights up to the 2^-24 weight quantum and w_l := 1 - w_r, and its total
    charge is exact."""
    L = 50.0
    for N, Ng, E_, bpe in ((200_000, 512, 2, 0), (30_001, 257, 1, 3), (4_000, 64, 3, 1)):
        rng = np.random.default_rng(N)
        x0 = rng.uniform(0, L, (E_, N)).astype(np.float32)
        x0[:, :4] = [0.0, np.nextafter(np.float32(L), np.float32(0)), L / Ng, 3 * L / Ng]
        v0 = rng.normal(0, 1, (E_, N)).astype(np.float32)
        dens = {}
        for acc in ("fixed", "fix64"):
            env = oc.BatchedPIC(E_, N, Ng, L=L, dt=0.05, dtype="float32", accum_dtype=acc, blocks_per_env=bpe)
            env.reset(x0, v0)
            n_reset, _, _ = env.fields()
            env.step(None, 3)
            n_step, E, _ = env.fields()
            dens[acc] = (n_reset, n_step, E, env.particles())
        dx = L / Ng
        per_particle = 1.0 * L / N / dx                     # density one whole particle adds to a node
        for k in (0, 1):
            d = np.abs(dens["fixed"][k] - dens["fix64"][k]).max()
            # float32 w_l + w_r misses 1 by up to ~ulp(x)/dx = 4e-5; measured 1.9e-4 of this unit at Ng = 257
            assert d < 1e-3 * per_particle * (N / Ng) ** 0.5, (N, k, d)
        assert np.abs(dens["fixed"][0].sum(axis=1) * dx - L).max() < 1e-12          # integer sums of w_l + w_r = 1: charge is exact
        assert rel_err(dens["fixed"][2], dens["fix64"][2]) < 1e-4                   # measured <= 1e-5
        assert np.abs(dens["fixed"][3][1] - dens["fix64"][3][1]).max() < 2e-5       # measured <= 1.5e-6


def test_fixed_point_accumulator_needs_float32_cic(oc):
    from ocplasma_amd._abi import PicError
    with pytest.raises(PicError, match="float32 particles"):
        oc.BatchedPIC(1, 1000, 64, dtype="float64", accum_dtype="fixed")
    with pytest.raises(PicError, match="float64 particles"):
        oc.BatchedPIC(1, 1000, 64, dtype="float32", accum_dtype="float64")
    with pytest.raises(PicError, match="CIC only"):
        oc.BatchedPIC(1, 1000, 64, dtype="float32", accum_dtype="fixed", interpol="TSC")
    env = oc.BatchedPIC(1, 1000, 64, dtype="float32", interpol="TSC")      # default falls back to the 64-bit fixed-point accumulator
    env.reset(np.zeros((1, 1000), np.float32) + 1.0, np.zeros((1, 1000), np.float32))
    assert abs(env.fields()[0].sum() * (50.0 / 64) - 50.0) < 1e-5              # float32 TSC weights sum to 1 within 4e-8


def test_torch_zero_copy_views(oc, po):
    import torch
    N, Ng, L = 10000, 128, 50.0
    x0, v0 = po.synthetic_bump_on_tail(N, L, seed=2)
    env = oc.BatchedPIC(3, N, Ng, L=L, dt=0.1)
    env.reset(np.stack([x0] * 3), np.stack([v0] * 3))
    env.step()
    env.sync()
    t = env.torch_views()
    x, v = env.particles()
    assert t["x"].is_cuda and t["x"].shape == (3, N)
    assert np.array_equal(t["x"].cpu().numpy(), x) and np.array_equal(t["v"].cpu().numpy(), v)
    assert np.array_equal(t["E_mesh"].cpu().numpy(), env.fields()[1])
    assert np.array_equal(t["PE_reward"].cpu().numpy(), env.energies()[2])
    obs = torch.cat([t["x"], t["v"]], dim=1)                 # the (2N,) observation, built on device
    assert np.array_equal(obs.cpu().numpy(), env.get_state())


def test_device_actuator_and_feedback_modes(oc, po):
    """actions -> E_ext on the device (golden g8 pins the host tables it uses) and the
    first Fourier modes of E_mesh (golden g9 pins the spectrum definition)."""
    E_, N, Ng, L, M = 3, 25000, 250, 50.0, 5
    xs, vs = zip(*[po.synthetic_two_stream(N, L, seed=70 + e) for e in range(E_)])
    x0, v0 = np.stack(xs), np.stack(vs)
    act = oc.E_field(L, Ng, M)
    a = oc.BatchedPIC(E_, N, Ng, L=L, dt=0.05)
    b = oc.BatchedPIC(E_, N, Ng, L=L, dt=0.05)
    a.set_actuator(act)
    for env in (a, b):
        env.reset(x0, v0)
    rng = np.random.default_rng(3)
    for k in range(4):
        actions = rng.uniform(-1.25, 1.25, (E_, 2 * M))
        a.step_actions(actions, nsteps=2)                        # E_ext built on the device
        b.step(act.compute_E_batched(actions), nsteps=2)         # E_ext built by the host mirror
    (xa, va), (xb, vb) = a.particles(), b.particles()
    assert circ_err(xa, xb, L) / L < 1e-13 and rel_err(va, vb) < 1e-12
    # modes of the current field vs numpy's FFT, definition of spectrum.py:16
    _, Em, _ = a.fields()
    ref = (np.fft.fft(Em, axis=1) / Ng * 2.0)[:, 1:M + 1]
    ek = a.modes(M)
    assert rel_err(ek, ref) < 1e-12
    fb = a.feedback_actions(M)
    assert fb.shape == (E_, 2 * M) and np.allclose(fb[:, :M], -ref.real, atol=1e-14) and np.allclose(fb[:, M:], ref.imag, atol=1e-14)
    # the same numbers through the reference-shaped host function on a state snapshot
    st = a.get_state()[0].reshape(-1, 1)
    ks, Ek = oc.compute_E_k_spectrum(1.0, L, L / Ng, Ng, st, False)
    assert rel_err(Ek[1:M + 1, 0], ek[0]) < 1e-10


def test_run_wo_oc_shaped_driver():
    """The reference's baseline driver flow (run_wo_oc.py) end to end on the drop-in classes."""
    import importlib.util
    import os
    from conftest import ROOT
    spec = importlib.util.spec_from_file_location("run_wo_oc_example", os.path.join(ROOT, "examples", "run_wo_oc.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    np.random.seed(42)
    out = mod.main(["--simcase", "bump-on-tail", "--num_particle", "10000", "--num_mesh", "128", "--t_max", "5"], quiet=True)
    assert out["snapshot"].shape == (20000, 50) and out["E"].shape == (50,)
    # same seed and sampler stream as the golden config-1 trajectory -> same energies step by step
    g = load_golden("g5_bump_on_tail_N10000_Ng128")
    assert rel_err(out["E"], g["H"][1:51]) < 1e-10 and rel_err(out["PE"], g["PE"][1:51]) < 1e-8
    assert rel_err(out["J_ee"], g["PE_reward"][1:51]) < 1e-8
    assert abs(out["J_KL"][0]) < 1.0 and np.isfinite(out["J_KL"]).all()


def test_device_sampler_is_keyed_by_the_global_environment_index(oc):
    """env_index_base: a handle holding environments [3, 5) of an ensemble draws exactly what a handle holding all of
    them draws for those environments (what ShardedPIC relies on)."""
    whole = oc.BatchedPIC(5, 3000, 64)
    part = oc.BatchedPIC(2, 3000, 64, env_index_base=3)
    for env in (whole, part):
        env.reset_sampled("bump-on-tail", seed=77)
    xw, vw = whole.particles()
    xp, vp = part.particles()
    assert np.array_equal(xw[3:], xp) and np.array_equal(vw[3:], vp)
    assert not np.array_equal(xw[0], xw[3])
    whole.close()
    part.close()


def test_sharded_rollout_example_two_ranks_one_device(tmp_path):
    """examples/sharded_rollout.py under torch.distributed.run with 2 ranks (gloo, both on cuda:0): the gathered
    returns and final energies are BIT FOR BIT the single-process ones -- the ensemble depends neither on the number
    of ranks nor on how many environments share a handle ("8-rank == 1-rank env-by-env, bitwise").  That
    holds because every sum on the path is an integer sum."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    script = os.path.join(ROOT, "examples", "sharded_rollout.py")
    common = ["--envs", "6", "--particles", "20000", "--mesh", "64", "--steps", "8", "--backend", "gloo", "--same-device"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")

    def run(cmd, out):
        r = subprocess.run(cmd + ["--out", str(out)], capture_output=True, text=True, timeout=300, env=env)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        return np.load(out)

    one = run([sys.executable, script] + common, tmp_path / "one.npz")
    two = run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
               "--master-addr", "127.0.0.1", "--master-port", "29533", script] + common, tmp_path / "two.npz")
    assert one["returns"].shape == (6,)
    assert np.array_equal(one["returns"], two["returns"])
    # KE is a float64 sum over workgroups (fixed order per geometry; here the geometry is the same), PE / PE_reward
    # come from integer deposits: all three agree exactly
    assert np.array_equal(one["energies"], two["energies"])
    assert (one["returns"] > 0).all() and (one["returns"] <= 8).all()


def test_steps_are_bitwise_reproducible_and_geometry_independent(oc, po):
    """Integer deposit sums (LDS and global) make a run independent of the order in which atomics land: the same
    100 steps twice give identical bits, and so does a different number of workgroups per environment (positions,
    velocities, density, field; KE is summed per workgroup in float64 and may differ in the last bits there)."""
    import torch
    E_, N, Ng, L = 8, 1_000_000, 256, 50.0

    def run(bpe, accum=None, steps=100):
        env = oc.BatchedPIC(E_, N, Ng, L=L, dt=0.1, blocks_per_env=bpe, accum_dtype=accum)
        env.reset_sampled("bump-on-tail", seed=11)
        env.step(None, steps)
        t = env.torch_views()
        env.sync()
        out = (t["x"].clone(), t["v"].clone(), env.fields(), env.energies())
        env.close()
        return out

    a, b, c = run(0), run(0), run(37)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    assert all(np.array_equal(p, q) for p, q in zip(a[2], b[2])) and all(np.array_equal(p, q) for p, q in zip(a[3], b[3]))
    assert torch.equal(a[0], c[0]) and torch.equal(a[1], c[1])
    assert all(np.array_equal(p, q) for p, q in zip(a[2], c[2]))
    assert np.array_equal(a[3][1], c[3][1]) and np.allclose(a[3][0], c[3][0], rtol=1e-14)
    # the float64 LDS accumulator (ds_add_f64) gives the same physics to rounding, not the same bits
    d = run(0, "float64", steps=10)
    e = run(0, None, steps=10)
    assert float((d[0] - e[0]).abs().max()) < 1e-11 and rel_err(d[2][1], e[2][1]) < 1e-10
    record_measure("determinism.f64acc_vs_fix64.x_10_steps", float((d[0] - e[0]).abs().max()))


def test_external_write_through_views_needs_invalidate(oc, po):
    """The handle caches the next step's first deposit; a caller that writes x / v through the zero-copy views
    (re-seeding environments on the device) must call invalidate() or refresh() -- then the step is the oracle's."""
    import torch
    E_, N, Ng, L = 2, 50_000, 128, 50.0
    xs, vs = zip(*[po.synthetic_bump_on_tail(N, L, seed=60 + e) for e in range(E_)])
    xn, vn = po.synthetic_two_stream(N, L, seed=77)
    for how in ("invalidate", "refresh"):
        env = oc.BatchedPIC(E_, N, Ng, L=L, dt=0.1)
        env.reset(np.stack(xs), np.stack(vs))
        env.step(None, 2)
        env.sync()
        t = env.torch_views()
        t["x"][1].copy_(torch.as_tensor(xn, device="cuda:0"))        # environment 1 re-seeded on the device
        t["v"][1].copy_(torch.as_tensor(vn, device="cuda:0"))
        torch.cuda.synchronize()
        getattr(env, how)()
        env.step()
        x, v = env.particles()
        ref = po.OraclePIC(xn, vn, Ng, L=L, dt=0.1, perturb=False, faithful=False)
        ref.update_state(None)
        assert circ_err(x[1], ref.x, L) / L < 1e-13 and rel_err(v[1], ref.v) < 1e-13, how
        assert rel_err(env.fields()[1][1], ref.E_mesh) < 1e-10
        env.close()


def test_resets_and_probes_inside_a_staged_step(oc, po):
    """A device-sampled reset abandons an open staged step (as pic_reset does); probes (eval_field / compute_E /
    solve_poisson) have their own accumulator, so they may run between stages without disturbing the step."""
    N, Ng, L = 20_000, 128, 50.0
    x0, v0 = po.synthetic_bump_on_tail(N, L, seed=8)
    h = oc._abi.Handle(N, Ng, 1, L, 1.0, 0.1)
    h.reset(x0[None], v0[None])
    h.step_stage(1)
    h.reset_sampled("two-stream", seed=3)          # abandons the staged step
    h.step(None, 1)                                # not refused
    h.reset(x0[None], v0[None])
    ref = po.OraclePIC(x0, v0, Ng, L=L, dt=0.1, perturb=False, faithful=False)
    ref.update_state(None)
    xp = np.random.default_rng(0).uniform(0, L, (1, N))
    n_alone, E_alone, _ = h.eval_field(xp)
    for stage in (1, 2, 3):
        h.step_stage(stage)
        n_mid, E_mid, _ = h.eval_field(xp)         # a probe in the middle of the step
        assert np.array_equal(n_mid, n_alone) and np.array_equal(E_mid, E_alone)
        h.solve_poisson(n_alone - 1.0)
    x, v = h.particles()
    assert circ_err(x[0], ref.x, L) / L < 1e-13 and rel_err(v[0], ref.v) < 1e-13
    assert rel_err(h.fields()[1][0], ref.E_mesh) < 1e-10
    h.close()


def test_feedback_control_loop_on_device():
    """run_feedback.py-shaped closed loop through modes -> action -> device actuator -> step: the
    controlled two-stream plasma must stay far below the free one's saturated field energy."""
    import importlib.util
    import os
    from conftest import ROOT
    spec = importlib.util.spec_from_file_location("feedback_control", os.path.join(ROOT, "examples", "feedback_control.py"))
    fc = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fc)
    pf, pc, effort = fc.run(num_envs=2, N=20000, steps=300, verbose=False)
    assert pf[-75:].mean() > 2.0                       # the instability did develop without control
    assert pc[-75:].mean() < 0.2 * pf[-75:].mean()     # and was suppressed with it
    assert effort.max() < 1.25                         # inside the trainers' action range (|a| <= 1.25)


@pytest.mark.parametrize("which", ["side", "default"])
def test_stream_ordered_torch_loop_matches_host_loop(oc, po, which):
    """The environment on torch's stream: modes -> action (torch ops) -> step, 30 iterations with no host
    synchronisation inside the loop, must reproduce the host-synchronous loop -- on a side stream and on torch's
    default stream (hipStream_t 0: a stream like any other, not "none")."""
    import contextlib
    import torch
    E_, N, Ng, L, M = 3, 20000, 128, 50.0, 4
    xs, vs = zip(*[po.synthetic_two_stream(N, L, seed=90 + e) for e in range(E_)])
    x0, v0 = np.stack(xs), np.stack(vs)
    act = oc.E_field(L, Ng, M)
    host = oc.BatchedPIC(E_, N, Ng, L=L, dt=0.1)
    dev = oc.BatchedPIC(E_, N, Ng, L=L, dt=0.1)
    for env in (host, dev):
        env.set_actuator(act)
        env.reset(x0, v0)
    side = torch.cuda.Stream()
    ctx = torch.cuda.stream(side) if which == "side" else contextlib.nullcontext()
    with ctx:                                           # the stream shared with the library
        dev.use_torch_stream()
        assert (torch.cuda.current_stream().cuda_stream == 0) == (which == "default")
        gain = torch.full((1, 2 * M), 0.8, dtype=torch.float64, device="cuda:0")
        total = torch.zeros(E_, dtype=torch.float64, device="cuda:0")
        for _ in range(30):
            a = dev.feedback_actions_torch(M) * gain    # torch kernels and library kernels interleave on one stream
            dev.step_actions_torch(a)
            total += dev.rewards_torch()
        torch.cuda.current_stream().synchronize()
    ret = np.zeros(E_)
    for _ in range(30):
        host.step_actions(host.feedback_actions(M) * 0.8)
        ret += host.rewards()
    (xh, vh), (xd, vd) = host.particles(), dev.particles()
    assert np.array_equal(xh, xd) and np.array_equal(vd, vh)         # same integer deposits: same bits
    assert np.allclose(total.cpu().numpy(), ret, rtol=1e-13)
    dev.use_own_stream()
    dev.step()                                          # still usable on its own stream afterwards
    dev.sync()


def test_rewards_torch_is_ordered_behind_the_step_on_the_handles_own_stream(oc, po):
    """The handle steps on its own stream; rewards_torch() / trainer_rewards_torch() read a zero-copy view on torch's.  Without
    a shared stream they must order the read themselves (no manual sync by the caller)."""
    E_, N, Ng, L = 4, 200_000, 128, 50.0
    env = oc.BatchedPIC(E_, N, Ng, L=L, dt=0.1)
    env.reset_sampled("two-stream", seed=4)
    for _ in range(5):
        env.step(None, 3)                                    # asynchronous: ~0.1 ms of queued work behind this call
        r = env.rewards_torch().cpu().numpy()
        tr = env.trainer_rewards_torch(alpha=2.0, beta=0.0).cpu().numpy()
        per = env.energies()[2]
        assert np.array_equal(r, np.maximum(1.0 - per, 0.0)) and np.array_equal(tr, 2.0 * np.maximum(1.0 - per, 0.0))
    env.close()


def test_phase_histogram_and_kl_on_device(oc):
    """estimate_f / compute_kl_divergence with the histogram counted on the device; the
    golden states include values on interior edges, on both outer edges and outside the range."""
    g = load_golden("g11_phase_hist")
    nb, L = int(g["nbins"]), float(g["L"])
    Np = g["st1"].shape[0] // 2
    env = oc.BatchedPIC(3, Np, 64, L=L, dt=0.1)
    xs = np.stack([g["st" + k][:Np, 0] for k in ("0", "1", "2")])
    vs = np.stack([g["st" + k][Np:, 0] for k in ("0", "1", "2")])
    env._h.set_particles(xs, vs)               # keep the raw values (reset would wrap x = L to 0)
    f = env.phase_density(nb, -25.0, 25.0)
    for e, k in enumerate(("0", "1", "2")):
        assert np.array_equal(f[e], g["f" + k]), k
    kl = env.kl_divergence(g["f0"], -25.0, 25.0)
    assert np.allclose(kl, [float(g["kl0"]), float(g["kl1"]), float(g["kl2"])], rtol=1e-12, atol=1e-15)


def test_edge_sizes(oc, po):
    """Ragged and extreme shapes: fewer particles than one tile, one particle, the largest mesh,
    a mesh over the LDS limit, zero steps.
    Parity is checked only where the reference itself is defined: for some (L, Ng) pairs the Sherman-Morrison denominator of
    its periodic solve (src/env/solve.py:27-53) is exactly 0 and its fields are inf / nan (the oracle, which restates
    that solve, warns "divide by zero" here).  For those shapes the oracle branch below is SKIPPED -- parity is undefined there,
    not green -- and only the size-independent properties (charge, no bad positions) are asserted; `skipped` names them."""
    L = 50.0
    skipped = []
    for N, Ng in ((1, 8), (7, 16), (511, 64), (513, 64), (4097, 2700)):
        rng = np.random.default_rng(N)
        x0, v0 = rng.uniform(0, L, N), rng.normal(0, 1, N)
        env = oc.BatchedPIC(2, N, Ng, L=L, dt=0.05)
        env.reset(np.stack([x0, x0]), np.stack([v0, -v0]))
        env.step(None, 0)
        env.step(None, 3)
        ref = po.OraclePIC(x0, v0, Ng, L=L, dt=0.05, perturb=False, faithful=False)
        assert ref.dt == env.dt
        if np.isfinite(ref.E_mesh).all():       # the reference's own solve blows up for some (L, Ng) pairs
            for _ in range(3):
                ref.update_state(None)
            x, v = env.particles()
            assert circ_err(x[0], ref.x, L) / L < 1e-12 and rel_err(v[0], ref.v) < 1e-11, (N, Ng)
            assert rel_err(env.fields()[1][0], ref.E_mesh) < 1e-9, (N, Ng)
        else:
            skipped.append((N, Ng))
        n = env.fields()[0]
        assert np.allclose(n.sum(axis=1) * (L / Ng), L, rtol=1e-12) and env.bad_count() == 0
        env.close()
    assert len(skipped) < 5, skipped              # at least one of the shapes is compared with the oracle
    print("test_edge_sizes: the reference's solve is singular, no parity asserted, for (N, Ng) in", skipped)
    with pytest.raises(oc._abi.PicError, match="Ng too large"):
        oc.BatchedPIC(1, 1000, 4096, L=L, dt=0.05)


@pytest.mark.parametrize("dtype,ng_sweeps,ng_resident", [("float64", 2722, 1159), ("float32", 3267, 1249)])
def test_largest_meshes_are_admitted_at_create_or_refused_there(oc, dtype, ng_sweeps, ng_resident):
    """The LDS budget of a workgroup (64 KB: dynamic meshes + the kernels' static arrays) is checked by pic_create: the largest
    admitted mesh of either schedule steps, one cell more is EINVAL at create -- never a launch failure later -- and the
    message quotes the bound that is enforced."""
    L, N = 50.0, 3000
    rng = np.random.default_rng(ng_sweeps)
    x0, v0 = rng.uniform(0, L, (1, N)).astype(dtype), rng.normal(0, 1, (1, N)).astype(dtype)
    x0[x0 >= L] = 0.0
    for Ng, bpe in ((ng_sweeps, 2), (ng_resident, -1)):
        env = oc.BatchedPIC(1, N, Ng, L=L, dt=0.05, dtype=dtype, blocks_per_env=bpe)
        env.reset(x0, v0)
        env.step(None, 2)
        n = env.fields()[0]
        assert np.allclose(n.sum() * (L / Ng), L, rtol=1e-6 if dtype == "float32" else 1e-12) and env.bad_count() == 0
        env.close()
    with pytest.raises(oc._abi.PicError, match=f"at most {ng_sweeps} cells"):
        oc.BatchedPIC(1, N, ng_sweeps + 1, L=L, dt=0.05, dtype=dtype, blocks_per_env=2)
    with pytest.raises(oc._abi.PicError, match=f"Ng <= {ng_resident} "):
        oc.BatchedPIC(1, N, ng_resident + 1, L=L, dt=0.05, dtype=dtype, blocks_per_env=-1)
    env = oc.BatchedPIC(1, N, ng_resident + 1, L=L, dt=0.05, dtype=dtype)       # automatic choice: the sweeps take it
    assert env._h.schedule() == "streaming"
    env.close()


def test_many_small_envs(oc, po):
    """1024 environments in one handle (BASELINE config 5's env count at a small N)."""
    E_, N, Ng, L = 1024, 2000, 64, 50.0
    rng = np.random.default_rng(0)
    x0, v0 = rng.uniform(0, L, (E_, N)), rng.normal(0, 1, (E_, N))
    env = oc.BatchedPIC(E_, N, Ng, L=L, dt=0.05)
    env.reset(x0, v0)
    env.step(None, 5)
    ke, pe, per = env.energies()
    for e in (0, 511, 1023):
        ref = po.OraclePIC(x0[e], v0[e], Ng, L=L, dt=0.05, perturb=False, faithful=False)
        for _ in range(5):
            ref.update_state(None)
        assert abs(ke[e] / ref.kinetic_energy() - 1) < 1e-12 and abs(pe[e] / ref.get_electric_energy() - 1) < 1e-8
    assert env.bad_count() == 0


def test_device_sampler_matches_reference_distributions(oc):
    """pic_reset_sampled draws the reference's TwoStream / BumpOnTail distributions on the device.  It is a
    different RNG, so the check is statistical: two-sample Kolmogorov-Smirnov against the host samplers
    (which reproduce the reference's stream, golden g10), moments, ordering, reproducibility."""
    from scipy import stats
    N, Ng, L = 200_000, 128, 50.0
    env = oc.BatchedPIC(3, N, Ng, L=L, dt=0.1)
    np.random.seed(123)
    for kind, host in (("bump-on-tail", oc.BumpOnTail(a=0.2, v0=3.0, sigma=1.0, n_samples=N, L=L)),
                       ("two-stream", oc.TwoStream(v0=3.0, sigma=1.0, n_samples=N, L=L))):
        env.reset_sampled(kind, a=0.2, v0=3.0, sigma=1.0, A=0.0, n_mode=2, seed=7)       # A = 0: raw sample
        x, v = env.particles()
        xh, vh = host.get_sample()
        assert (x >= 0).all() and (x < L).all() and (np.abs(v) <= 10).all() and env.bad_count() == 0
        n1 = int(N * (1 / 1.2)) if kind == "bump-on-tail" else N // 2
        for sl in (slice(0, n1), slice(n1, N)):                                          # per population
            assert stats.ks_2samp(v[0][sl], vh[sl]).pvalue > 1e-3, kind
        assert stats.ks_2samp(x[0], xh).pvalue > 1e-3
        assert stats.kstest(x[1] / L, "uniform").pvalue > 1e-3
        if kind == "bump-on-tail":
            assert abs(v[0][:n1].mean()) < 0.02 and abs(v[0][n1:].mean() - 3.0) < 0.03
            assert abs(v[0][:n1].std() - 1.0) < 0.02
        else:
            assert abs(v[0][:n1].mean() - 3.0) < 0.02 and abs(v[0][n1:].mean() + 3.0) < 0.02
        assert not np.array_equal(x[0], x[1]) and abs(np.corrcoef(x[0], x[1])[0, 1]) < 0.02   # envs differ
        env.reset_sampled(kind, a=0.2, v0=3.0, sigma=1.0, A=0.0, n_mode=2, seed=7)
        x2, v2 = env.particles()
        assert np.array_equal(x, x2) and np.array_equal(v, v2)                           # same seed, same sample
        env.reset_sampled(kind, a=0.2, v0=3.0, sigma=1.0, A=0.1, n_mode=2, seed=7)       # perturbation (pic.py:68)
        x3, v3 = env.particles()
        assert np.array_equal(x3, x) and np.allclose(v3, v * (1 + 0.1 * np.sin(2 * np.pi * 2 * x / L)), rtol=1e-13)
        env.reset_sampled(kind, a=0.2, v0=3.0, sigma=1.0, A=0.1, n_mode=2, seed=8)
        assert not np.array_equal(env.particles()[0], x)
    # the sampled state is a valid starting point: fields are there and stepping conserves energy
    ke0, pe0, _ = env.energies()
    env.step(None, 20)
    ke, pe, _ = env.energies()
    assert np.max(np.abs((ke + pe) / (ke0 + pe0) - 1)) < 1e-3 and env.bad_count() == 0


def test_pic_api_corners(oc, po):
    """Less-travelled parts of the drop-in surface: update_params(dt), assigning x / v, float32 state,
    update_density / update_E_field, and ShardedPIC on a single rank with the real BatchedPIC behind it."""
    g = load_golden("g4_bump_on_tail_ext_N4000_Ng256")
    L = float(g["L"])
    sim = make_pic(oc, g)
    # update_params(dt=...) re-creates the device handle and carries the particles over
    x_before = sim.x.copy()
    sim.update_params(dt=0.05, gamma=None)
    assert sim.dt == 0.05 and np.array_equal(sim.x, x_before)
    sim.update_state(None)
    ref = po.OraclePIC(g["x0_raw"], g["v0_raw"], int(g["Ng"]), L=L, dt=0.05, A=float(g["A"]), n_mode=int(g["n_mode"]),
                       perturb=True, faithful=False)
    ref.update_state(None)
    assert circ_err(sim.x, ref.x, L) / L < 1e-13 and rel_err(sim.v, ref.v) < 1e-13
    # assigning particles uploads them and refreshes the fields
    sim.x = ref.x[::-1].copy()
    sim.v = ref.v[::-1].copy()
    assert rel_err(sim.E_mesh, ref.E_mesh) < 1e-11 and abs(sim.get_energy() / ref.get_energy() - 1) < 1e-13
    sim.update_density()
    sim.update_E_field()
    assert rel_err(sim.n, ref.n) < 1e-12
    # compute_state_gradient (pic.py:125-129) on an arbitrary state, with and without an external field
    eta = np.concatenate([ref.x + 0.3, 1.7 * ref.v])
    for ext in (None, 0.1 * np.random.default_rng(1).normal(size=(int(g["Ng"]), 1))):
        ours = sim.compute_state_gradient(eta.copy(), ext)
        theirs = ref.state_gradient(eta.copy(), ext)
        assert ours.shape == theirs.shape == (2 * sim.N, 1)
        assert np.array_equal(ours[:sim.N], theirs[:sim.N]) and rel_err(ours[sim.N:], theirs[sim.N:]) < 1e-11
    sim.close()
    # float32 drop-in: the API still hands out float64 arrays
    s32 = make_pic(oc, g, dtype="float32")
    s32.update_state(None)
    assert s32.x.dtype == np.float64 and s32.get_state().shape == (8000, 1)
    assert abs(s32.get_energy() / float(g["H"][0]) - 1) < 1e-3
    s32.close()
    # ... and with fixed-point positions (the start is handed over as float32, which bounds what one step can agree to)
    sfx = oc.PIC(N=int(g["N"]), N_mesh=int(g["Ng"]), n0=float(g["n0"]), L=L, dt=float(g["dt_in"]), gamma=float(g["gamma"]),
                 A=float(g["A"]), n_mode=int(g["n_mode"]), init_dist=FixedDist(g["x0_raw"], g["v0_raw"]), dtype="float32",
                 position_dtype="fixed32")
    sfx.update_state(None)
    ref1 = po.OraclePIC(g["x0_raw"], g["v0_raw"], int(g["Ng"]), L=L, dt=float(g["dt_in"]), A=float(g["A"]), n_mode=int(g["n_mode"]),
                        perturb=True, faithful=False)
    ref1.update_state(None)
    assert circ_err(sfx.x, ref1.x, L) / L < 3e-7 and rel_err(sfx.v, ref1.v) < 2e-6 and rel_err(sfx.E_mesh, ref1.E_mesh) < 1e-4
    sfx.close()
    # one-rank ShardedPIC = plain BatchedPIC
    sh = oc.env.sharded.ShardedPIC(3, 5000, 64, L=L, dt=0.1)
    xs, vs = zip(*[po.synthetic_two_stream(5000, L, seed=300 + e) for e in range(3)])
    sh.reset(np.stack(xs), np.stack(vs), is_global=True)
    sh.step(None, 3)
    r = sh.gather_returns()
    e = sh.gather_energies()
    assert r.shape == (3,) and e.shape == (3, 3) and (r >= 0).all() and (r <= 1).all()
    sh.close()


def test_gym_step_takes_actions_through_the_device_actuator(oc):
    """PIC.step(action): the coefficient vector a policy emits -> E_field on the device -> update_state; same
    trajectory as handing update_state the host mirror's compute_E (golden g4 actions)."""
    g = load_golden("g4_bump_on_tail_ext_N4000_Ng256")
    L, Ng = float(g["L"]), int(g["Ng"])
    mm = g["actions"].shape[1] // 2
    a_env, b_env = make_pic(oc, g), make_pic(oc, g)
    act = oc.E_field(L, Ng, mm)
    a_env.set_actuator(act)
    for k in range(5):
        a = g["actions"][k]
        obs, reward, done, info = a_env.step(a)
        act.update_E(a[:mm], a[mm:])
        pe_pre = b_env.get_reward_electric_energy()
        b_env.update_state(E_external=act.compute_E())
        assert obs.shape == (2 * a_env.N, 1) and done is False
        assert abs(reward - max(1.0 - pe_pre, 0.0)) < 1e-12
        assert abs(info["PE"] / b_env.get_electric_energy() - 1) < 1e-10
    assert circ_err(a_env.x, b_env.x, L) / L < 1e-12 and rel_err(a_env.v, b_env.v) < 1e-12
    assert circ_err(a_env.x, g["x_1"], L) > 0        # it did move on from step 1
    a_env.close()
    b_env.close()


def test_update_state_w_input_func(oc, po):
    """pic.py:148-163: the external field is a function of the sub-stage state.  Checked against the oracle's
    Yoshida-4 composition driven by the same input function (3 steps)."""
    g = load_golden("g4_bump_on_tail_ext_N4000_Ng256")
    L, Ng, N = float(g["L"]), int(g["Ng"]), int(g["N"])
    mesh = np.linspace(0, L, Ng).reshape(-1, 1)
    calls = []

    def input_func(eta):                    # a pure function of the (unwrapped) sub-stage state
        calls.append(float(eta[:N].max()))
        a = np.mean(np.cos(2 * np.pi * eta[:N] / L))
        b = np.mean(eta[N:] ** 2)
        return 0.3 * a * np.sin(2 * np.pi * mesh / L) + 0.02 * b * np.cos(4 * np.pi * mesh / L)

    sim = make_pic(oc, g)
    ref = po.OraclePIC(g["x0_raw"], g["v0_raw"], Ng, L=L, dt=float(g["dt_in"]), A=float(g["A"]), n_mode=int(g["n_mode"]),
                       perturb=True, faithful=True)
    assert sim.dt == ref.dt
    for _ in range(3):
        sim.update_state_w_input_func(input_func)
        eta = np.concatenate([ref.x.reshape(-1, 1), ref.v.reshape(-1, 1)], axis=0)
        eta = po.yoshida4(eta, lambda z: ref.state_gradient(z, input_func(z)), ref.dt)
        ref.x, ref.v = np.mod(eta[:N], L), eta[N:]
        ref.update_density()
        ref.update_E_field()
    assert circ_err(sim.x, ref.x, L) / L < 1e-13 and rel_err(sim.v, ref.v) < 1e-12
    assert rel_err(sim.E_mesh, ref.E_mesh) < 1e-10 and abs(sim.get_energy() / ref.get_energy() - 1) < 1e-13
    # None = plain update_state; the staged entry point enforces its order and blocks pic_step mid-step
    sim.update_state_w_input_func(None)
    ref.update_state(None)
    assert circ_err(sim.x, ref.x, L) / L < 1e-13
    h = sim._ensure_handle()
    with pytest.raises(oc._abi.PicError, match="order"):
        h.step_stage(2)
    h.step_stage(1)
    with pytest.raises(oc._abi.PicError, match="staged step"):
        h.step(None, 1)
    h.step_stage(2)
    h.step_stage(3)
    h.step(None, 1)
    sim.close()


def test_create_destroy_does_not_leak(oc, po):
    import torch
    N, Ng, L = 200_000, 256, 50.0
    x0, v0 = po.synthetic_bump_on_tail(N, L, seed=1)

    def cycle():
        env = oc.BatchedPIC(4, N, Ng, L=L, dt=0.1)
        env.set_actuator(oc.E_field(L, Ng, 3))
        env.reset(np.stack([x0] * 4), np.stack([v0] * 4))
        env.step_actions(np.zeros((4, 6)), 2)
        env.modes(3)
        env.eval_field(np.stack([x0] * 4))
        env.close()

    cycle()
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    for _ in range(25):
        cycle()
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info()[0]
    assert free0 - free1 < 64 << 20, (free0, free1)      # 25 cycles of ~40 MB each would show up as ~1 GB


def test_placement_search_frees_what_it_does_not_keep(oc):
    """States of 256 MB and more: pic_create allocates blocks until x and v stream well together and must give
    back every block but the two it keeps, on success and when the handle is destroyed."""
    import torch
    oc.BatchedPIC(1, 1000, 32, L=50.0, dt=0.1).close()              # (the HIP runtime's one-time allocations behind the first handle
    torch.cuda.synchronize()                                        # of a process, ~146 MB, are not this test's subject)
    free0 = torch.cuda.mem_get_info()[0]
    state = 2 * 6 * 3_000_000 * 8                                   # 288 MB of float64 x and v
    for _ in range(3):
        env = oc.BatchedPIC(6, 3_000_000, 128, L=50.0, dt=0.1)
        tried, kept, slowest, secs = env._h.placement_info()
        assert tried >= 1 and kept >= slowest > 0.0 and secs < 0.5
        env.sync()
        held = free0 - torch.cuda.mem_get_info()[0]
        assert state <= held < state + (96 << 20), (held, state)    # particles + meshes + staging, nothing of the search
        env.reset_sampled("two-stream", seed=5)
        env.step(None, 2)
        assert env.bad_count() == 0
        env.close()
    torch.cuda.synchronize()
    assert free0 - torch.cuda.mem_get_info()[0] < 32 << 20


def test_placement_search_resumes_at_resets_until_the_addresses_are_out(oc):
    """The search for an (x, v) placement runs in legs: pic_create runs one; while it has ended only for lack of time, every reset --
    which replaces the particles anyway -- runs another (at most four in all) and may move v; once pic_device_ptrs has handed the
    addresses out, v stays where it is.  A 1 ms budget per leg (pic_config.placement_ms) makes every leg run out of time; what the
    handle computes is the same bits wherever v lies, and nothing of the search stays allocated."""
    import torch
    E_, N, Ng, L = 6, 3_000_000, 128, 50.0
    oc.BatchedPIC(1, 1000, 32, L=L, dt=0.1).close()
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    ref = oc.BatchedPIC(E_, N, Ng, L=L, dt=0.1, placement="off")
    ref.reset_sampled("bump-on-tail", seed=7)
    ref.step(None, 3)
    want = ref.particles() + ref.fields()
    ref.close()
    env = oc.BatchedPIC(E_, N, Ng, L=L, dt=0.1, placement_ms=1)
    st = env._h.placement_stats()
    assert st["legs"] == 1 and st["outcome"] in ("timeout", "found")     # ("found": a fast pair among the very first blocks)
    legs = [st["legs"]]
    for k in range(5):                                                   # resets run further legs, four in all at most
        env.reset_sampled("bump-on-tail", seed=7)
        st = env._h.placement_stats()
        legs.append(st["legs"])
        assert st["legs"] <= 4
        if st["outcome"] != "timeout":
            break
    assert legs == sorted(legs)
    if st["outcome"] == "timeout":
        assert legs[-1] == 4 and legs[-2] == 4                           # the fifth reset ran none
    env.step(None, 3)
    got = env.particles() + env.fields()
    assert all(np.array_equal(p, q) for p, q in zip(got, want))
    held = free0 - torch.cuda.mem_get_info()[0]
    state = 2 * E_ * N * 8
    assert state <= held < state + (96 << 20), (held, state)             # x, v, meshes, staging: nothing of any leg
    # a handle whose addresses are out keeps its v
    other = oc.BatchedPIC(E_, N, Ng, L=L, dt=0.1, placement_ms=1)
    ptr_v = other._h.device_ptrs()["v"]
    before = other._h.placement_stats()["legs"]
    other.reset_sampled("bump-on-tail", seed=7)
    assert other._h.placement_stats()["legs"] == before and other._h.device_ptrs()["v"] == ptr_v
    other.step(None, 3)
    assert all(np.array_equal(p, q) for p, q in zip(other.particles() + other.fields(), want))
    other.close()
    env.close()
    with pytest.raises(oc._abi.PicError, match="placement_ms"):
        oc.BatchedPIC(1, 1000, 32, L=L, dt=0.1, placement_ms=-1)


def test_placement_off_and_tight_memory(oc):
    """pic_config.placement = off: no search, nothing timed, x | v in one block -- and the same bits (where the particles lie
    is not arithmetic).  With little more than the state itself free on the device the search has no room and a handle must
    still come up."""
    import torch
    E_, N, Ng, L = 6, 3_000_000, 128, 50.0
    runs = {}
    for mode in ("auto", "off"):
        env = oc.BatchedPIC(E_, N, Ng, L=L, dt=0.1, placement=mode)
        info = env._h.placement_info()
        assert (info[0] >= 1 and info[1] > 0.0) if mode == "auto" else info == (1, 0.0, 0.0, 0.0)
        env.reset_sampled("bump-on-tail", seed=3)
        env.step(None, 3)
        t = env.torch_views()
        env.sync()
        runs[mode] = (t["x"].clone(), t["v"].clone(), env.fields())
        env.close()
    assert torch.equal(runs["auto"][0], runs["off"][0]) and torch.equal(runs["auto"][1], runs["off"][1])
    assert all(np.array_equal(p, q) for p, q in zip(runs["auto"][2], runs["off"][2]))
    # leave about 2.2 x the particle state free (the search may hold a third of what is free: less than one more block)
    state = 2 * E_ * N * 8
    torch.cuda.synchronize()
    torch.cuda.empty_cache()                              # (what torch caches would be handed out again without touching the device)
    hog = []
    for _ in range(200):                                  # in pieces: one 280 GB block is a lot to ask of any allocator
        left = int(torch.cuda.mem_get_info()[0] - 2.2 * state)
        if left < (64 << 20):
            break
        hog.append(torch.empty(min(left, 8 << 30), dtype=torch.uint8, device="cuda:0"))
    assert torch.cuda.mem_get_info()[0] < 3 * state
    env = oc.BatchedPIC(E_, N, Ng, L=L, dt=0.1)
    env.reset_sampled("bump-on-tail", seed=3)
    env.step(None, 3)
    t = env.torch_views()
    env.sync()
    assert torch.equal(t["x"], runs["off"][0]) and env.bad_count() == 0
    env.close()
    del hog
    torch.cuda.empty_cache()


def test_two_handles_from_two_threads(oc, po):
    """Different handles are independent (own stream, own buffers); ctypes drops the GIL during calls."""
    import threading
    N, Ng, L = 50_000, 128, 50.0
    inputs = [po.synthetic_two_stream(N, L, seed=200 + k) for k in range(2)]
    out = [None, None]

    def work(k):
        env = oc.BatchedPIC(2, N, Ng, L=L, dt=0.05)
        x, v = inputs[k]
        env.reset(np.stack([x, x]), np.stack([v, v]))
        for _ in range(40):
            env.step()
        out[k] = env.particles()
        env.close()

    ts = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    for k in range(2):
        env = oc.BatchedPIC(2, N, Ng, L=L, dt=0.05)
        x, v = inputs[k]
        env.reset(np.stack([x, x]), np.stack([v, v]))
        env.step(None, 40)
        xs, vs = env.particles()
        assert circ_err(out[k][0], xs, L) / L < 1e-10 and rel_err(out[k][1], vs) < 1e-9
        env.close()


def test_errors_cross_the_abi_as_codes(oc):
    h = oc._abi.Handle(1000, 64, 1, 50.0, 1.0, 0.1)
    with pytest.raises(oc._abi.PicError, match="pic_reset first"):
        h.step()
    h.reset(np.linspace(0, 49, 1000)[None], np.zeros((1, 1000)))
    # a NaN position is counted, not followed into memory
    x = np.linspace(0, 49, 1000)
    x[5] = np.nan
    h.reset(x[None], np.zeros((1, 1000)))
    assert h.bad_count() >= 1
    h.close()


def test_fixed_point_format_through_the_whole_api(oc, po):
    """position_dtype="fixed32": every entry point that takes or returns positions converts at the boundary
    (float32 in, float32 out) and agrees with the float64 handle to what float32 inputs allow."""
    E_, N, Ng, L = 2, 30_000, 128, 50.0
    xs, vs = zip(*[po.synthetic_bump_on_tail(N, L, seed=700 + e) for e in range(E_)])
    x0, v0 = np.stack(xs).astype(np.float32), np.stack(vs).astype(np.float32)
    x0[x0 >= L] = 0.0
    fx = oc.BatchedPIC(E_, N, Ng, L=L, dt=0.1, dtype="float32", position_dtype="fixed32")
    hi = oc.BatchedPIC(E_, N, Ng, L=L, dt=0.1)
    fx.reset(x0, v0)
    hi.reset(x0.astype(np.float64), v0.astype(np.float64))
    # particles come back as float32 within half a float32 ulp of what went in (the fixed-point grid is 327 x finer)
    xb, vb = fx.particles()
    assert xb.dtype == np.float32 and np.array_equal(vb, v0) and circ_err(xb, x0, L) <= 2e-6
    assert rel_err(fx.fields()[1], hi.fields()[1]) < 1e-6 and rel_err(fx.fields()[0], hi.fields()[0]) < 1e-6
    # probes on arbitrary positions (compute_E), the gather at the particles, CIC bookkeeping
    xp = np.random.default_rng(1).uniform(-10, 60, (E_, N)).astype(np.float32)          # also outside the box
    nf, Ef, pf = fx.eval_field(xp)
    nh, Eh, ph = hi.eval_field(xp.astype(np.float64))
    assert rel_err(nf, nh) < 1e-6 and rel_err(Ef, Eh) < 1e-5 and np.allclose(pf, ph, rtol=1e-5)
    assert rel_err(fx.gather_E(), hi.gather_E()) < 1e-5
    jl, jr, wl, wr = fx._h.cic(1)
    jl2, jr2, wl2, wr2 = hi._h.cic(1)
    same = jl == jl2                                     # a particle within 1e-8 of a cell edge may fall on either side
    assert same.mean() > 0.9999 and np.array_equal(jr[same], jr2[same]) and np.max(np.abs(wr[same] - wr2[same])) < 1e-6
    assert np.allclose(wl + wr, 1.0, atol=1e-7)
    # phase-space histogram and Fourier modes of the current state
    cf, ch = fx._h.phase_histogram(32, -10.0, 10.0), hi._h.phase_histogram(32, -10.0, 10.0)
    assert np.abs(cf.astype(np.int64) - ch.astype(np.int64)).sum() <= 4 and cf.sum() == ch.sum()
    fx.step(None, 5)
    hi.step(None, 5)
    assert rel_err(fx.modes(3), hi.modes(3)) < 1e-4
    # zero-copy views: raw uint32 positions and their float64 image
    fx.sync()
    t = fx.torch_views()
    assert str(t["x_fixed"].dtype) == "torch.int32" and t["x"].dtype.is_floating_point
    assert circ_err(t["x"].cpu().numpy(), fx.particles()[0], L) <= 2e-6
    assert fx.bad_count() == 0
    fx.close()
    hi.close()


def test_trainer_reward_on_device_matches_reward_class(oc, po):
    """BatchedPIC.trainer_rewards_torch = Reward.compute_reward (reward.py:71-76) per environment, without the host
    round trip and the second deposit the reference's reward needs (golden g4 pins Reward itself)."""
    import torch
    E_, N, Ng, L, M = 4, 20000, 128, 50.0, 5
    xs, vs = zip(*[po.synthetic_two_stream(N, L, seed=800 + e) for e in range(E_)])
    env = oc.BatchedPIC(E_, N, Ng, L=L, dt=0.1)
    env.reset(np.stack(xs), np.stack(vs))
    env.step(None, 30)
    env.sync()
    actions = np.random.default_rng(4).uniform(-1.25, 1.25, (E_, 2 * M))
    got = env.trainer_rewards_torch(torch.as_tensor(actions, device="cuda:0"), alpha=0.7, beta=1.3).cpu().numpy()
    x, v = env.particles()
    for e in range(E_):
        st = np.concatenate([x[e], v[e]]).reshape(-1, 1)
        rew = oc.Reward(st, Ng, L, -25.0, 25.0, 1.0, 0.7, 1.3, n_actions=2 * M)
        assert abs(got[e] - rew.compute_reward(st, actions[e])) < 1e-9          # Reward re-deposits the host state
        assert abs(got[e] - rew.reward_from_energy(env.energies()[2][e], actions[e])) < 1e-13
    assert np.allclose(env.trainer_rewards_torch().cpu().numpy(), env.rewards(), rtol=1e-14)
    env.close()
