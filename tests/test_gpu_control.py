"""Controlled rollouts in one call (round 3): pic_step_actions_traj / pic_step_ext_traj / pic_step_feedback and the
resident schedule's launch-to-launch hand-over of the next step's first deposit.

What they stand for in the reference: the trainers' inner loop, one new action per step
(src/control/rl/ddpg.py:421-468 -> E_field.update_E / compute_E, src/control/actuator.py:46-63 -> PIC.update_state,
src/env/pic.py:131-146), PIC.simulate(E_external_traj) (pic.py:175-223) and the feedback loop of run_feedback.py:130-168.
Pinned by the g4 golden trajectories (20 steps, a new action each) and, bit for bit, by the step-by-step host loops the
earlier rounds pinned against the oracle."""
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


def _state(env):
    x, v = env.particles()
    n, E, phi = env.fields()
    return x, v, n, E, phi


def _same_bits(a, b):
    return all(np.array_equal(p, q) for p, q in zip(_state(a), _state(b)))


@pytest.mark.parametrize("name,interpol", [("g4_bump_on_tail_ext_N4000_Ng256", "CIC"),
                                           ("g4_two_stream_ext_N3000_Ng200", "CIC"),
                                           ("g4_tsc_bump_on_tail_ext_N3000_Ng128", "TSC")])
@pytest.mark.parametrize("bpe", [-1, 2])
def test_g4_action_trajectory_in_one_call(oc, name, interpol, bpe):
    """The reference's 20 controlled steps (a new action every step) as ONE pic_step_actions_traj call, on both schedules,
    at the tolerances test_g4_external_field_steps holds for the step-by-step loop."""
    g = load_golden(name)
    L, Ng, N = float(g["L"]), int(g["Ng"]), int(g["N"])
    mm = g["actions"].shape[1] // 2
    env = oc.BatchedPIC(1, N, Ng, L=L, dt=float(g["dt"]), interpol=interpol, blocks_per_env=bpe)
    assert env._h.schedule() == ("resident" if bpe < 0 else "streaming")
    env.set_actuator(oc.E_field(L, Ng, mm))
    env.reset(g["x_init"].reshape(1, N), g["v_init"].reshape(1, N))
    ke0, pe0, per0 = env.energies()
    ke, pe, per = env.step_actions_traj(g["actions"][:, None, :], history=True)
    x, v, n, E, _ = _state(env)
    ex, ev = circ_err(x[0], g["x_20"], L) / L, rel_err(v[0], g["v_20"])
    eE, eH = rel_err(E[0], g["E_mesh_20"]), rel_err(np.concatenate([ke0 + pe0, (ke + pe)[:, 0]]), g["H"])
    record_measure(f"{name}.traj.bpe{bpe}.x_20", ex)
    record_measure(f"{name}.traj.bpe{bpe}.E_mesh_20", eE)
    assert ex < 1e-10 and ev < 1e-10 and eE < 1e-9 and eH < 1e-12
    assert rel_err(n[0], g["n_20"]) < 1e-10
    if interpol == "CIC":          # the reward of step k is taken on the state BEFORE it (ddpg.py:455): PE_reward[k-1]
        pre = np.concatenate([per0, per[:-1, 0]])
        assert rel_err(pre, g["PE_reward"]) < 1e-10
    env.close()


@pytest.mark.parametrize("N,Ng,bpe,dtype,pos", [(5000, 250, -1, "float64", None), (30000, 128, 0, "float64", None),
                                               (5000, 250, -1, "float32", "fixed32"), (30000, 128, 3, "float32", None)])
def test_one_call_rollouts_equal_the_step_by_step_loops(oc, po, N, Ng, bpe, dtype, pos):
    """actions_traj, ext_traj and feedback in ONE call each against the loops of single calls they replace: same bits in
    particles and fields, same recorded energies and actions."""
    E_, L, M, K = 3, 50.0, 3, 7
    rng = np.random.default_rng(N + Ng)
    xs, vs = zip(*[po.synthetic_two_stream(N, L, seed=20 + e) for e in range(E_)])
    x0, v0 = np.stack(xs).astype(dtype), np.stack(vs).astype(dtype)
    x0[x0 >= L] = 0.0
    act = oc.E_field(L, Ng, M)
    kw = dict(L=L, dt=0.1, blocks_per_env=bpe, dtype=dtype, position_dtype=pos)

    def fresh():
        env = oc.BatchedPIC(E_, N, Ng, **kw)
        env.set_actuator(act)
        env.reset(x0, v0)
        return env

    actions = rng.uniform(-1.25, 1.25, (K, E_, 2 * M))
    # (1) a new action every step
    a, b = fresh(), fresh()
    ke, pe, per = a.step_actions_traj(actions, history=True)
    loop = []
    for k in range(K):
        b.step_actions(actions[k])
        loop.append(b.energies())
    assert _same_bits(a, b)
    assert np.array_equal(pe, np.stack([e[1] for e in loop])) and np.array_equal(per, np.stack([e[2] for e in loop]))
    assert np.allclose(ke, np.stack([e[0] for e in loop]), rtol=1e-14)
    # ... continued without history (asynchronous form), then a plain step: the cached deposits stay consistent
    a.step_actions_traj(actions[:3])
    a.step()
    for k in range(3):
        b.step_actions(actions[k])
    b.step()
    assert _same_bits(a, b)
    # (2) a new mesh field every step
    fields = 0.05 * rng.normal(size=(K, E_, Ng))
    ke, pe, per = a.step_ext_traj(fields, history=True)
    for k in range(K):
        b.step(fields[k])
    assert _same_bits(a, b) and np.array_equal(pe[-1], b.energies()[1])
    # (3) the feedback law, K steps in one call
    rec = a.step_feedback(K, actions=True, history=True)
    acts = []
    for k in range(K):
        acts.append(b.feedback_actions(M))
        b.step_actions(acts[-1])
    assert np.array_equal(rec["actions"], np.stack(acts))
    assert _same_bits(a, b) and np.array_equal(rec["PE_reward"][-1], b.energies()[2])
    a.close()
    b.close()


def test_resident_calls_hand_the_next_deposit_over(oc, po):
    """Single-step calls of the resident schedule take the LDS mesh of the next step's first deposit over from the call before
    (no entry deposit, particles loaded under the first field phase) -- and, in float64 handles of at most 32 environments like
    this one, the cell and weights every particle's first position was located with: same bits as one multi-step call and as the streaming
    sweeps, also across everything that must drop the hand-over (invalidate after a write through the views, set_particles,
    a staged step)."""
    import torch
    E_, N, Ng, L = 3, 5000, 250, 50.0
    xs, vs = zip(*[po.synthetic_bump_on_tail(N, L, seed=40 + e) for e in range(E_)])
    x0, v0 = np.stack(xs), np.stack(vs)
    one = oc.BatchedPIC(E_, N, Ng, L=L, dt=0.1)
    many = oc.BatchedPIC(E_, N, Ng, L=L, dt=0.1)
    seq = oc.BatchedPIC(E_, N, Ng, L=L, dt=0.1, blocks_per_env=2)
    assert one._h.schedule() == "resident" and seq._h.schedule() == "streaming"
    ext = 0.03 * np.random.default_rng(5).normal(size=(E_, Ng))
    for env in (one, many, seq):
        env.reset(x0, v0)
    for k in range(6):
        one.step(ext if k % 2 else None)
    many.step(None); many.step(ext); many.step(None, nsteps=1); many.step(ext); many.step(None); many.step(ext)
    for k in range(6):
        seq.step(ext if k % 2 else None)
    assert _same_bits(one, seq) and _same_bits(many, seq)
    # a write through the zero-copy views: the mesh handed over belongs to the old particles
    for env in (one, seq):
        t = env.torch_views()
        env.sync()
        t["v"].mul_(1.01)
        torch.cuda.synchronize()
        env.invalidate()
        env.step(None, 2)
    assert _same_bits(one, seq)
    # set_particles (no refresh) and a staged step drop it too
    x, v = seq.particles()
    for env in (one, seq):
        env._h.set_particles(x[:, ::-1].copy(), v[:, ::-1].copy())
        env.step(ext)
        env._h.step_stage(1, ext); env._h.step_stage(2, None); env._h.step_stage(3, ext)
        env.step(None)
    assert _same_bits(one, seq)
    ref = po.OraclePIC(x0[0], v0[0], Ng, L=L, dt=0.1, perturb=False, faithful=False)
    for k in range(6):
        ref.update_state(ext[0].reshape(-1, 1) if k % 2 else None)
    chk = oc.BatchedPIC(1, N, Ng, L=L, dt=0.1)
    chk.reset(x0[:1], v0[:1])
    for k in range(6):
        chk.step(ext[:1] if k % 2 else None)
    xc, vc = chk.particles()
    assert circ_err(xc[0], ref.x, L) / L < 1e-13 and rel_err(vc[0], ref.v) < 1e-12
    for env in (one, many, seq, chk):
        env.close()


def test_pic_simulate_with_a_field_trajectory(oc, po):
    """PIC.simulate(E_external_traj) (pic.py:175-223) through pic_step_ext_traj: snapshots and energy traces equal the
    update_state loop's."""
    np.random.seed(7)
    N, Ng, L = 3000, 128, 50.0
    mk = lambda: oc.PIC(N=N, N_mesh=Ng, n0=1.0, L=L, dt=0.1, tmin=0.0, tmax=1.2, A=0.1, n_mode=2,
                        init_dist=oc.TwoStream(v0=3.0, sigma=1.0, n_samples=N, L=L))
    sim = mk()
    x0, v0 = sim.x.copy(), sim.v.copy()
    Nt = int(np.ceil((sim.tmax - sim.tmin) / sim.dt))
    traj = [0.04 * np.random.default_rng(k).normal(size=(Ng, 1)) for k in range(Nt)]
    snap, Es, PEs = sim.simulate(traj)
    assert snap.shape == (2 * N, Nt + 1) and Es.shape == (Nt + 1,) and PEs.shape == (Nt + 1,)
    h = oc.BatchedPIC(1, N, Ng, L=L, dt=sim.dt)
    h.reset(x0.reshape(1, N), v0.reshape(1, N))
    for k in range(Nt):
        h.step(traj[k].reshape(1, Ng))
        x, v = h.particles()
        assert np.array_equal(snap[:N, k + 1], x[0]) and np.array_equal(snap[N:, k + 1], v[0])
        ke, pe, _ = h.energies()
        assert PEs[k + 1] == pe[0] and abs(Es[k + 1] / (ke[0] + pe[0]) - 1) < 1e-14
    assert np.array_equal(sim.x[:, 0], x[0])
    h.close()
    sim.close()


@pytest.mark.parametrize("bpe", [-1, 2])
def test_field_trajectory_with_snapshots_on_both_schedules(oc, po, bpe):
    """pic_step_ext_traj with the particle snapshots PIC.simulate records (pic.py:175-223), resident (the kernel writes them from
    its registers) and streaming (a copy kernel after every step): snapshots, energies and the final state equal the loop of
    single steps."""
    E_, N, Ng, L, K = 2, 4000, 128, 50.0, 5
    xs, vs = zip(*[po.synthetic_two_stream(N, L, seed=60 + e) for e in range(E_)])
    x0, v0 = np.stack(xs), np.stack(vs)
    fields = 0.05 * np.random.default_rng(8).normal(size=(K, E_, Ng))
    a = oc.BatchedPIC(E_, N, Ng, L=L, dt=0.1, blocks_per_env=bpe)
    b = oc.BatchedPIC(E_, N, Ng, L=L, dt=0.1, blocks_per_env=bpe)
    for env in (a, b):
        env.reset(x0, v0)
    sx, sv, ke, pe, per = a.step_ext_traj(fields, snapshots=True)
    for k in range(K):
        b.step(fields[k])
        x, v = b.particles()
        assert np.array_equal(sx[k], x) and np.array_equal(sv[k], v)
        kb, pb, rb = b.energies()
        assert np.array_equal(pe[k], pb) and np.array_equal(per[k], rb) and np.allclose(ke[k], kb, rtol=1e-14)
    assert _same_bits(a, b)
    a.close()
    b.close()


def test_sub_rows_of_the_accumulators_do_not_change_a_bit(oc, po):
    """Few large environments spread every accumulator row over several sub-rows (pic_device.h: acc_row_sum); a handle with
    many workgroups per environment (sub-rows) and one with few (a single row) must agree bit for bit, and with the oracle."""
    N, Ng, L = 300_000, 256, 50.0
    x0, v0 = po.synthetic_bump_on_tail(N, L, seed=9)
    ext = 0.02 * np.random.default_rng(1).normal(size=(1, Ng))
    a = oc.BatchedPIC(1, N, Ng, L=L, dt=0.05)                        # automatic: 74 workgroups, 8 sub-rows
    b = oc.BatchedPIC(1, N, Ng, L=L, dt=0.05, blocks_per_env=5)      # 5 workgroups: one row
    ref = po.OraclePIC(x0, v0, Ng, L=L, dt=0.05, perturb=False, faithful=False)
    for env in (a, b):
        env.reset(x0[None], v0[None])
    for k in range(4):
        for env in (a, b):
            env.step(ext if k % 2 else None)
        ref.update_state(ext[0].reshape(-1, 1) if k % 2 else None)
    assert _same_bits(a, b)
    x, v = a.particles()
    n, E, _ = a.fields()
    assert np.array_equal(a.eval_field(x)[0], n)                      # probes go through the sub-rows as well
    assert circ_err(x[0], ref.x, L) / L < 1e-13 and rel_err(v[0], ref.v) < 1e-12 and rel_err(E[0], ref.E_mesh) < 1e-10
    a.close()
    b.close()


@pytest.mark.parametrize("N,Ng,bpe,dtype,pos", [(5000, 250, 0, "float64", "float"), (5000, 250, 3, "float64", "float"),
                                               (3000, 96, 0, "float32", "fixed32"), (300_000, 128, 0, "float64", "float")])
def test_step_observe_is_the_step_and_its_getters_in_one_call(oc, po, N, Ng, bpe, dtype, pos):
    """pic_step_observe = update_state + get_state + the energies of the new state in one call with one synchronisation
    (ddpg.py:421-468): bit for bit what pic_step / pic_step_actions followed by pic_get_particles / pic_get_energies return, under
    a mesh field, under an action and free, on both schedules, through the pinned staging buffer (small states), the
    conversion kernel of the fixed-point positions and the pageable path (a state above 4 MB)."""
    L, E_ = 50.0, 2
    rng = np.random.default_rng(4)
    x0 = np.stack([po.synthetic_bump_on_tail(N, L, seed=s)[0] for s in range(E_)])
    v0 = np.stack([po.synthetic_bump_on_tail(N, L, seed=s)[1] for s in range(E_)])
    act = oc.E_field(L, Ng, 3)
    a = oc.BatchedPIC(E_, N, Ng, L=L, dt=0.1, dtype=dtype, position_dtype=pos, blocks_per_env=bpe)
    b = oc.BatchedPIC(E_, N, Ng, L=L, dt=0.1, dtype=dtype, position_dtype=pos, blocks_per_env=bpe)
    for env in (a, b):
        env.set_actuator(act)
        env.reset(x0, v0)
    for k in range(6):
        ext = 0.05 * rng.normal(size=(E_, Ng)) if k % 3 == 1 else None
        action = rng.uniform(-1.25, 1.25, (E_, 6)) if k % 3 == 2 else None
        state, (ke, pe, per) = a.step_observe(ext, action, nsteps=1 + k % 2)
        if action is not None:
            b.step_actions(action, nsteps=1 + k % 2)
        else:
            b.step(ext, nsteps=1 + k % 2)
        assert np.array_equal(state, b.get_state()), k
        for got, want in zip((ke, pe, per), b.energies()):
            assert np.array_equal(got, want), k
    assert _same_bits(a, b)
    with pytest.raises(oc._abi.PicError, match="alternatives"):
        a._h.step_observe(np.zeros((E_, Ng)), np.zeros((E_, 6)))
    a.close()
    b.close()
