"""The reference's own LOOPS around the step, against fixtures the reference itself produced (tests/golden/make_golden.py,
composed_loops: G12-G16, round 4):

  G12  run_feedback.py:130-168        the closed feedback loop (two-stream, N = 5000, Ng = 250, max_mode = 5, 50 steps)
  G13  src/env/pic.py:175-223         PIC.simulate with and without a field trajectory
  G14  src/control/rl/ddpg.py:364-381 the behaviour-cloning rollout with its hard-coded spectrum arguments (n0 = 1, L = 50, Ng = 250)
  G15  src/env/pic.py:148-163         update_state_w_input_func with a pure function of the sub-stage state
  G16  src/env/pic.py:125-129, 84-91  compute_state_gradient on arbitrary states; reinit() and the step after it

Each loop is run (a) as the reference writes it, with this package's objects in the reference's places, and (b) through the one-call
entry points that replace it on the device (pic_step_feedback, pic_step_ext_traj, pic_step_snapshots), on both schedules.  Bounds
are those of the g4 trajectories (tests/test_gpu_parity.py::test_g4_external_field_steps): positions 1e-10 of L, velocities 1e-10,
mesh field 1e-9, energies 1e-12 -- what was measured is recorded (profiles/r4_measured_errors.json)."""
import numpy as np
import pytest

from conftest import circ_err, load_golden, record_measure, rel_err

pytestmark = pytest.mark.gpu

TOL_X, TOL_V, TOL_E, TOL_H = 1e-10, 1e-10, 1e-9, 1e-12


@pytest.fixture(scope="module")
def oc():
    import ocplasma_amd
    return ocplasma_amd


def _check_marks(g, tag, k, x, v, E_mesh, L, n=None):
    ex, ev = circ_err(x, g[f"x_{k}"], L) / L, rel_err(v, g[f"v_{k}"])
    eE = rel_err(E_mesh, g[f"E_mesh_{k}"])
    record_measure(f"{tag}.x_{k}", ex)
    record_measure(f"{tag}.v_{k}", ev)
    record_measure(f"{tag}.E_mesh_{k}", eE)
    assert ex < TOL_X and ev < TOL_V and eE < TOL_E, (tag, k, ex, ev, eE)
    if n is not None:
        assert rel_err(n, g[f"n_{k}"]) < 1e-10


# ---------------------------------------------------------------------------------------------------------------------
# G12: run_feedback.py:130-168
# ---------------------------------------------------------------------------------------------------------------------
def test_g12_feedback_loop_as_the_reference_writes_it(oc):
    """run_feedback.py's loop body verbatim, with this package's PIC / E_field / Reward / compute_E_k_spectrum in the
    reference's places (every compute_E inside them runs on the device): per-step coefficients, energies, costs and the
    reward list, particles and mesh field after 1, 10 and 50 steps."""
    g = load_golden("g12_feedback_two_stream_N5000_Ng250")
    L, Ng, N, mm = float(g["L"]), int(g["Ng"]), int(g["N"]), int(g["max_mode"])
    np.random.seed(47)                                   # the fixture's seed: the samplers draw the reference's particles (g10)
    sim = oc.PIC(N=N, N_mesh=Ng, n0=1.0, L=L, dt=0.1, tmin=0.0, tmax=50.0, gamma=5.0, A=0.1, n_mode=2, interpol="CIC",
                 init_dist=oc.TwoStream(v0=3.0, sigma=1.0, n_samples=N, L=L))
    assert np.array_equal(sim.x, g["x_init"]) and np.array_equal(sim.v, g["v_init"])
    assert np.array_equal(sim.init_dist.get_init_state(), g["init_state"])
    actuator = oc.E_field(L, Ng, mm)
    reward = oc.Reward(sim.init_dist.get_init_state(), Ng, L, -25.0, 25.0, 1.0, 1.0, 1.0)
    worst = dict(coeff=0.0, H=0.0, PE=0.0, reward=0.0, ee=0.0, ie=0.0, kl=0.0)
    scale = np.max(np.abs(np.concatenate([g["coeff_cos"], g["coeff_sin"]], axis=1)))
    for k in range(1, 51):
        _, Eks = oc.compute_E_k_spectrum(1.0, L, L / Ng, Ng, sim.get_state(), False)
        Eks = Eks[1:mm + 1, :]
        actuator.update_E((-1) * np.real(Eks), (+1) * np.imag(Eks))
        coeffs = np.concatenate([actuator.coeff_cos.ravel(), actuator.coeff_sin.ravel()])
        E_external = actuator.compute_E()
        sim.update_state(E_external)
        want = np.concatenate([g["coeff_cos"][k - 1], g["coeff_sin"][k - 1]])
        worst["coeff"] = max(worst["coeff"], float(np.max(np.abs(coeffs - want)) / scale))
        assert rel_err(E_external[:, 0], g["E_external"][k - 1]) < 1e-9 or np.max(np.abs(g["E_external"][k - 1])) < 1e-12
        worst["H"] = max(worst["H"], abs(sim.get_energy() / g["H"][k - 1] - 1))
        worst["PE"] = max(worst["PE"], abs(sim.get_electric_energy() / g["PE"][k - 1] - 1))
        state = sim.get_state()
        worst["kl"] = max(worst["kl"], abs(reward.compute_kl_divergence(state) - g["cost_kl"][k - 1]))
        worst["ee"] = max(worst["ee"], abs(reward.compute_electric_energy(state) / g["cost_ee"][k - 1] - 1))
        worst["ie"] = max(worst["ie"], abs(reward.compute_input_energy(coeffs) - g["cost_ie"][k - 1]))
        worst["reward"] = max(worst["reward"], abs(reward.compute_reward(state, E_external) - g["reward"][k - 1]))
        if k in (1, 10, 50):
            _check_marks(g, "g12.host_loop", k, sim.x, sim.v, sim.E_mesh, L, sim.n)
    for name, val in worst.items():
        record_measure(f"g12.host_loop.{name}", val)
    assert worst["coeff"] < 1e-9 and worst["H"] < TOL_H and worst["PE"] < 1e-9 and worst["ee"] < 1e-9
    assert worst["ie"] < 1e-10 and worst["reward"] < 1e-9 and worst["kl"] < 1e-9
    sim.close()


@pytest.mark.parametrize("bpe", [-1, 2])
@pytest.mark.parametrize("one_call", [True, False])
def test_g12_feedback_loop_on_the_device(oc, bpe, one_call):
    """The same 50 steps without the host in the loop: pic_step_feedback (the law evaluated inside the field phases) in
    calls of 1 + 9 + 40 steps, and the host loop pic_get_modes -> pic_step_actions it replaces; resident and streaming."""
    g = load_golden("g12_feedback_two_stream_N5000_Ng250")
    L, Ng, N, mm = float(g["L"]), int(g["Ng"]), int(g["N"]), int(g["max_mode"])
    env = oc.BatchedPIC(1, N, Ng, L=L, dt=float(g["dt"]), blocks_per_env=bpe)
    assert env._h.schedule() == ("resident" if bpe < 0 else "streaming")
    env.set_actuator(oc.E_field(L, Ng, mm))
    env.reset(g["x_init"].reshape(1, N), g["v_init"].reshape(1, N))
    want_actions = np.concatenate([g["coeff_cos"], g["coeff_sin"]], axis=1)
    scale = np.max(np.abs(want_actions))
    tag = f"g12.{'one_call' if one_call else 'modes_loop'}.bpe{bpe}"
    acts, H, PE, PEr = [], [], [], []
    done = 0
    for upto in (1, 10, 50):
        if one_call:
            rec = env.step_feedback(upto - done, actions=True, history=True)
            acts.append(rec["actions"][:, 0, :])
            H.append((rec["KE"] + rec["PE"])[:, 0]); PE.append(rec["PE"][:, 0]); PEr.append(rec["PE_reward"][:, 0])
        else:
            for _ in range(upto - done):
                a = env.feedback_actions(mm)
                env.step_actions(a)
                ke, pe, per = env.energies()
                acts.append(a); H.append(ke + pe); PE.append(pe); PEr.append(per)
        done = upto
        x, v = env.particles()
        n, E, _ = env.fields()
        _check_marks(g, tag, upto, x[0], v[0], E[0], L, n[0])
    acts, H, PE, PEr = np.concatenate(acts), np.concatenate(H), np.concatenate(PE), np.concatenate(PEr)
    e_act = float(np.max(np.abs(acts - want_actions)) / scale)
    record_measure(f"{tag}.actions", e_act)
    assert e_act < 1e-9
    assert rel_err(H, g["H"]) < TOL_H and rel_err(PE, g["PE"]) < 1e-9
    assert rel_err(PEr, g["cost_ee"]) < 1e-9                      # Reward.compute_electric_energy of the post-step state
    assert env.bad_count() == 0
    env.close()


# ---------------------------------------------------------------------------------------------------------------------
# G13: PIC.simulate
# ---------------------------------------------------------------------------------------------------------------------
def _check_simulation(tag, snap, E, PE, g, prefix, L, N):
    want = g[prefix + "snapshot"]
    assert snap.shape == want.shape and E.shape == g[prefix + "E"].shape and PE.shape == g[prefix + "PE"].shape
    ex = max(circ_err(snap[:N, k], want[:N, k], L) for k in range(want.shape[1])) / L
    ev = rel_err(snap[N:], want[N:])
    record_measure(f"{tag}.x", ex)
    record_measure(f"{tag}.v", ev)
    assert np.array_equal(snap[:, 0], want[:, 0])                 # the initial column is the state handed over
    assert ex < TOL_X and ev < TOL_V
    assert rel_err(E, g[prefix + "E"]) < TOL_H and rel_err(PE, g[prefix + "PE"]) < 1e-9


def test_g13_simulate_with_a_field_trajectory(oc):
    """PIC.simulate(E_external_traj) (pic.py:175-223): snapshot (2N, Nt + 1) with the initial column, E and PE traces."""
    g = load_golden("g13_simulate")
    L, Ng, N, Nt = float(g["L"]), int(g["Ng"]), int(g["N"]), int(g["Nt"])
    np.random.seed(48)
    sim = oc.PIC(N=N, N_mesh=Ng, n0=1.0, L=L, dt=0.1, tmin=0.0, tmax=float(g["tmax"]), gamma=5.0, A=0.1, n_mode=2,
                 interpol="CIC", init_dist=oc.TwoStream(v0=3.0, sigma=1.0, n_samples=N, L=L))
    assert np.array_equal(sim.x, g["x_init"]) and np.array_equal(sim.v, g["v_init"])
    traj = [row.reshape(-1, 1) for row in g["E_external_traj"]]
    assert len(traj) == Nt
    snap, E, PE = sim.simulate(traj)
    _check_simulation("g13.PIC.simulate", snap, E, PE, g, "", L, N)
    assert rel_err(sim.E_mesh, g["E_mesh_final"]) < TOL_E
    sim.close()


@pytest.mark.parametrize("bpe", [-1, 2])
def test_g13_field_trajectory_in_one_call_on_both_schedules(oc, bpe):
    """pic_step_ext_traj with snapshots (what PIC.simulate runs on) straight through the batched handle."""
    g = load_golden("g13_simulate")
    L, Ng, N, Nt = float(g["L"]), int(g["Ng"]), int(g["N"]), int(g["Nt"])
    env = oc.BatchedPIC(1, N, Ng, L=L, dt=float(g["dt"]), blocks_per_env=bpe)
    env.reset(g["x_init"].reshape(1, N), g["v_init"].reshape(1, N))
    ke0, pe0, _ = env.energies()
    sx, sv, ke, pe, _ = env.step_ext_traj(g["E_external_traj"][:, None, :], snapshots=True)
    snap = np.concatenate([np.concatenate([g["x_init"], sx[:, 0, :].T], axis=1),
                           np.concatenate([g["v_init"], sv[:, 0, :].T], axis=1)], axis=0)
    E = np.concatenate([ke0 + pe0, (ke + pe)[:, 0]])
    PE = np.concatenate([pe0, pe[:, 0]])
    _check_simulation(f"g13.ext_traj.bpe{bpe}", snap, E, PE, g, "", L, N)
    env.close()


@pytest.mark.parametrize("bpe", [None, -1, 2])
def test_g13_free_simulation_tsc(oc, bpe):
    """PIC.simulate(None) with the TSC shape (bump-on-tail, N = 2000, Ng = 64): through the drop-in class (bpe None) and
    through pic_step_snapshots on both schedules."""
    g = load_golden("g13_simulate")
    L, Ng, N = float(g["L"]), int(g["free_Ng"]), int(g["free_N"])
    if bpe is None:
        np.random.seed(49)
        sim = oc.PIC(N=N, N_mesh=Ng, n0=1.0, L=L, dt=0.1, tmin=0.0, tmax=float(g["free_tmax"]), gamma=5.0, A=0.1, n_mode=2,
                     interpol="TSC", init_dist=oc.BumpOnTail(a=0.2, v0=3.0, sigma=1.0, n_samples=N, L=L))
        assert np.array_equal(sim.x, g["free_x_init"]) and np.array_equal(sim.v, g["free_v_init"])
        snap, E, PE = sim.simulate(None)
        sim.close()
    else:
        env = oc.BatchedPIC(1, N, Ng, L=L, dt=0.1, interpol="TSC", blocks_per_env=bpe)
        env.reset(g["free_x_init"].reshape(1, N), g["free_v_init"].reshape(1, N))
        ke0, pe0, _ = env.energies()
        Nt = g["free_snapshot"].shape[1] - 1
        sx, sv, (ke, pe, _) = _snapshots(env, Nt)
        snap = np.concatenate([np.concatenate([g["free_x_init"], sx[:, 0, :].T], axis=1),
                               np.concatenate([g["free_v_init"], sv[:, 0, :].T], axis=1)], axis=0)
        E, PE = np.concatenate([ke0 + pe0, (ke + pe)[:, 0]]), np.concatenate([pe0, pe[:, 0]])
        env.close()
    _check_simulation(f"g13.free_tsc.{'PIC' if bpe is None else 'bpe' + str(bpe)}", snap, E, PE, g, "free_", L, N)


def _snapshots(env, nsteps):
    out = env.simulate_snapshots(nsteps)
    return out[0], out[1], out[2:]


# ---------------------------------------------------------------------------------------------------------------------
# G14: the behaviour-cloning rollout of ddpg.py:364-381
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("tag", ["a", "b"])
def test_g14_bc_rollout_as_the_reference_writes_it(oc, tag):
    """ddpg.py:364-381 verbatim against this package's objects: the action comes from a spectrum evaluated with the HARD-CODED
    n0 = 1, L = 50, Ng = 250 whatever the environment's own mesh is (case b: Ng = 200), the reward is taken on the PRE-step
    state with the coefficient vector as its second argument (ddpg.py:381)."""
    g = load_golden("g14_bc_rollout")
    L, N, Ng, K = float(g["L"]), int(g[f"{tag}_N"]), int(g[f"{tag}_Ng"]), int(g[f"{tag}_steps"])
    np.random.seed({"a": 50, "b": 51}[tag])
    env = oc.PIC(N=N, N_mesh=Ng, n0=1.0, L=L, dt=0.1, tmin=0.0, tmax=50.0, gamma=5.0, A=0.1, n_mode=2, interpol="CIC",
                 init_dist=oc.BumpOnTail(a=0.2, v0=3.0, sigma=1.0, n_samples=N, L=L))
    assert np.array_equal(env.x, g[f"{tag}_x_init"]) and np.array_equal(env.v, g[f"{tag}_v_init"])
    actuator = oc.E_field(env.L, env.N_mesh, 5)
    reward_cls = oc.Reward(env.init_dist.get_init_state(), env.N_mesh, env.L, -25.0, 25.0, env.n0, 1.0, 0.5)
    max_mode = 5
    actions, rewards = [], []
    for idx_t in range(K):
        state = env.get_state()
        _, Eks = oc.compute_E_k_spectrum(1.0, 50.0, 50.0 / 250, 250, state, False)
        Eks = Eks[1:max_mode + 1, :]
        actuator.update_E((-1) * np.real(Eks), (+1) * np.imag(Eks))
        action = np.concatenate([actuator.coeff_cos.ravel(), actuator.coeff_sin.ravel()])
        env.update_state(E_external=actuator.compute_E())
        actions.append(action)
        rewards.append(reward_cls.compute_reward(state, action))
    actions, rewards = np.array(actions), np.array(rewards)
    e_act = float(np.max(np.abs(actions - g[f"{tag}_actions"])) / np.max(np.abs(g[f"{tag}_actions"])))
    e_rew = float(np.max(np.abs(rewards - g[f"{tag}_reward"])))
    ex = circ_err(env.x, g[f"{tag}_x_final"], L) / L
    ev, eE = rel_err(env.v, g[f"{tag}_v_final"]), rel_err(env.E_mesh, g[f"{tag}_E_mesh_final"])
    for name, val in (("actions", e_act), ("reward", e_rew), ("x", ex), ("v", ev), ("E_mesh", eE)):
        record_measure(f"g14{tag}.host_loop.{name}", val)
    assert e_act < 1e-9 and e_rew < 1e-9 and ex < TOL_X and ev < TOL_V and eE < TOL_E
    env.close()


@pytest.mark.parametrize("bpe", [-1, 2])
def test_g14_bc_rollout_on_the_device(oc, bpe):
    """Case a (the environment's mesh IS the hard-coded 250 on L = 50): the whole rollout is one pic_step_feedback call; its
    recorded actions are the behaviour-cloning targets, PE_reward of the state before each step the reward's energy term."""
    g = load_golden("g14_bc_rollout")
    L, N, Ng, K = float(g["L"]), int(g["a_N"]), int(g["a_Ng"]), int(g["a_steps"])
    env = oc.BatchedPIC(1, N, Ng, L=L, dt=float(g["a_dt"]), blocks_per_env=bpe)
    env.set_actuator(oc.E_field(L, Ng, 5))
    env.reset(g["a_x_init"].reshape(1, N), g["a_v_init"].reshape(1, N))
    per0 = env.energies()[2]
    rec = env.step_feedback(K, actions=True, history=True)
    acts = rec["actions"][:, 0, :]
    e_act = float(np.max(np.abs(acts - g["a_actions"])) / np.max(np.abs(g["a_actions"])))
    pre = np.concatenate([per0, rec["PE_reward"][:-1, 0]])           # the state BEFORE each step (ddpg.py:381)
    rew = oc.Reward(g["a_init_state"], Ng, L, -25.0, 25.0, 1.0, 1.0, 0.5)
    rewards = np.array([rew.reward_from_energy(pre[k], acts[k]) for k in range(K)])
    e_rew = float(np.max(np.abs(rewards - g["a_reward"])))
    x, v = env.particles()
    _, E, _ = env.fields()
    ex, ev, eE = circ_err(x[0], g["a_x_final"], L) / L, rel_err(v[0], g["a_v_final"]), rel_err(E[0], g["a_E_mesh_final"])
    for name, val in (("actions", e_act), ("reward", e_rew), ("x", ex), ("v", ev), ("E_mesh", eE)):
        record_measure(f"g14a.one_call.bpe{bpe}.{name}", val)
    assert e_act < 1e-9 and e_rew < 1e-9 and ex < TOL_X and ev < TOL_V and eE < TOL_E
    env.close()


# ---------------------------------------------------------------------------------------------------------------------
# G15: PIC.update_state_w_input_func (pic.py:148-163) with a pure input function
# ---------------------------------------------------------------------------------------------------------------------
def test_g15_update_state_w_input_func(oc):
    """The drop-in calls the input function 3 times per step where the reference calls it 7 times (first table):
    for a PURE function of the sub-stage state the step is the reference's -- particles, field and energy of 5 steps against the
    fixture, and the three states it hands to the function in step 1 are the reference's calls 1, 3 and 5 (the ones whose force
    the integrator uses, src/env/integration.py:31-32)."""
    g = load_golden("g15_input_func")
    L, Ng, N, K = float(g["L"]), int(g["Ng"]), int(g["N"]), int(g["steps"])
    np.random.seed(52)
    sim = oc.PIC(N=N, N_mesh=Ng, n0=1.0, L=L, dt=0.1, tmin=0.0, tmax=50.0, gamma=5.0, A=0.1, n_mode=2, interpol="CIC",
                 init_dist=oc.TwoStream(v0=3.0, sigma=1.0, n_samples=N, L=L))
    assert np.array_equal(sim.x, g["x_init"]) and np.array_equal(sim.v, g["v_init"])
    mesh = np.linspace(0, L, Ng).reshape(-1, 1)
    calls = []

    def input_func(eta):
        calls.append(eta.copy())
        a = np.mean(np.cos(2 * np.pi * eta[:N] / L))
        b = np.mean(eta[N:] ** 2)
        return 0.3 * a * np.sin(2 * np.pi * mesh / L) + 0.02 * b * np.cos(4 * np.pi * mesh / L)

    worst = dict(x=0.0, v=0.0, E_mesh=0.0, H=0.0, calls=0.0)
    for k in range(K):
        calls.clear()
        sim.update_state_w_input_func(input_func)
        assert len(calls) == 3 and int(g["calls_per_step"][k]) == 7
        if k == 0:
            for mine, theirs in zip(calls, g["useful_calls_step1"]):
                # sub-stage positions are unwrapped on both sides; compared on the circle in case a wrap differs by one box
                worst["calls"] = max(worst["calls"], circ_err(np.mod(mine[:N, 0], L), np.mod(theirs[:N], L), L) / L,
                                     rel_err(mine[N:, 0], theirs[N:]))
        worst["x"] = max(worst["x"], circ_err(sim.x, g["x"][k], L) / L)
        worst["v"] = max(worst["v"], rel_err(sim.v, g["v"][k]))
        worst["E_mesh"] = max(worst["E_mesh"], rel_err(sim.E_mesh, g["E_mesh"][k]))
        worst["H"] = max(worst["H"], abs(sim.get_energy() / g["H"][k] - 1))
    for name, val in worst.items():
        record_measure(f"g15.{name}", val)
    assert worst["calls"] < 1e-13 and worst["x"] < TOL_X and worst["v"] < TOL_V and worst["E_mesh"] < TOL_E and worst["H"] < TOL_H
    sim.close()


# ---------------------------------------------------------------------------------------------------------------------
# G16: PIC.compute_state_gradient on arbitrary states, and what reinit() leaves
# ---------------------------------------------------------------------------------------------------------------------
def test_g16_state_gradient_and_reinit(oc):
    """pic.py:125-129 with positions outside [0, L) -- the reference's compute_E wraps eta[:N] in place (util.py:51), and so does the
    drop-in's -- with and without an external field; then reinit() (pic.py:84-91): a fresh sample from the global RNG (the same
    draws as the reference's, g10), E / E_mesh / phi_mesh None until the next step, and that step against the reference's."""
    g = load_golden("g16_gradient_reinit")
    L, Ng, N = float(g["L"]), int(g["Ng"]), int(g["N"])
    np.random.seed(53)
    sim = oc.PIC(N=N, N_mesh=Ng, n0=1.0, L=L, dt=0.1, tmin=0.0, tmax=50.0, gamma=5.0, A=0.1, n_mode=2, interpol="CIC",
                 init_dist=oc.BumpOnTail(a=0.2, v0=3.0, sigma=1.0, n_samples=N, L=L))
    for ext, key in ((None, "grad_free"), (g["E_ext"], "grad_ext")):
        eta = g["eta"].copy()
        out = sim.compute_state_gradient(eta, ext)
        assert out.shape == (2 * N, 1) and np.array_equal(out[:N], g[key][:N])
        e = rel_err(out[N:], g[key][N:])
        record_measure(f"g16.{key}", e)
        assert e < 1e-11
        assert np.array_equal(eta, g["eta_after"])                      # wrapped in place, like the reference's
    sim.update_state(None)
    assert circ_err(sim.x, g["x_before"], L) / L < TOL_X and rel_err(sim.E_mesh, g["E_mesh_before"]) < TOL_E
    sim.reinit()
    assert np.array_equal(sim.x, g["x_reinit"]) and np.array_equal(sim.v, g["v_reinit"])
    assert sim.E is None and sim.E_mesh is None and sim.phi_mesh is None and bool(g["fields_none"].all())
    assert rel_err(sim.n, g["n_reinit"]) < 1e-13
    sim.update_state(None)
    ex, ev, eE = circ_err(sim.x, g["x_after"], L) / L, rel_err(sim.v, g["v_after"]), rel_err(sim.E_mesh, g["E_mesh_after"])
    for name, val in (("x_after", ex), ("v_after", ev), ("E_mesh_after", eE)):
        record_measure(f"g16.{name}", val)
    assert ex < TOL_X and ev < TOL_V and eE < TOL_E and abs(sim.get_energy() / float(g["H_after"]) - 1) < TOL_H
    sim.close()
