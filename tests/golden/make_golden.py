#!/usr/bin/env python3
"""Generate the golden vectors in this directory from the real reference.

Runs ONLY in the build container (needs ``/root/reference``); the ``.npz`` files
it writes are committed and are what travels to the GPU box.  Usage::

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

Caveat, stated wherever parity is claimed: ``numba`` is not installed in the
image and ``src/env/solve.py:3`` / ``src/env/util.py:2`` do ``from numba import
jit``.  This script puts a throw-away ``numba`` module whose ``jit`` returns
the undecorated function on ``sys.path`` (in a temp dir, never in the repo), so
the four decorated functions run as the plain Python/NumPy they are written as.
Everything else is the reference's own code, imported unmodified.

What is stored (SURVEY.md 8c, G1..G9 + sampler and TSC extras): inputs and the
reference's outputs only -- no reference source.
"""
import os
import sys
import tempfile

import numpy as np

REF = os.environ.get("PIC_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))


def _import_reference():
    shim = tempfile.mkdtemp(prefix="numba_identity_")
    os.makedirs(os.path.join(shim, "numba"))
    with open(os.path.join(shim, "numba", "__init__.py"), "w") as f:
        f.write("def jit(*a, **k):\n"
                "    if len(a) == 1 and callable(a[0]) and not k:\n"
                "        return a[0]\n"
                "    return lambda fn: fn\n")
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    sys.path.insert(0, shim)


def save(name, **arrays):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrays)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


def main():
    _import_reference()
    from src.env.pic import PIC
    from src.env.dist import BumpOnTail, TwoStream
    from src.env.interpolate import CIC, TSC
    from src.env.solve import Gaussian_Elimination_Periodic
    from src.env.util import (compute_E, compute_n, generate_grad, generate_laplacian,
                              compute_electric_energy, compute_hamiltonian)
    from src.control.actuator import E_field
    from src.control.rl.reward import Reward
    from src.interpret.spectrum import compute_E_k_spectrum

    rng = np.random.default_rng(20261004)

    # ---- G1: deposit (CIC + TSC) through compute_n, edge inputs included -----
    L, Ng, n0 = 50.0, 64, 1.0
    dx = L / Ng
    x = rng.uniform(-L, 2 * L, size=2000)
    x[:8] = [0.0, np.nextafter(L, 0.0), -1e-3, L + 1e-3, -1e-20, L, 3.5 * L, -2.25 * L]
    x[8:8 + Ng] = np.arange(Ng) * dx                 # exactly on the nodes
    x = x.reshape(-1, 1)
    N = x.shape[0]
    u = x.copy()
    n, jl, jr, wl, wr = compute_n(u, dx, Ng, n0, L, N, True, "CIC")
    ut = x.copy()
    nt, tl, tm, tr, twl, twm, twr = compute_n(ut, dx, Ng, n0, L, N, True, "TSC")
    # direct CIC (single mod) on in-range inputs
    xin = rng.uniform(0, L, size=(500, 1))
    n_d, jl_d, jr_d, wl_d, wr_d = CIC(xin.copy(), n0, L, 500, Ng, dx)
    save("g1_deposit", L=L, Ng=Ng, n0=n0, x=x, x_wrapped=u, n=n, jl=jl, jr=jr, wl=wl, wr=wr,
         tsc_n=nt, tsc_jl=tl, tsc_jm=tm, tsc_jr=tr, tsc_wl=twl, tsc_wm=twm, tsc_wr=twr,
         xin=xin, n_d=n_d, jl_d=jl_d, jr_d=jr_d, wl_d=wl_d, wr_d=wr_d)

    # ---- G2: n -> phi, E_mesh for several mesh sizes and gammas ---------------
    g2 = {}
    for Ngi in (128, 250, 256, 1024):
        dxi = L / Ngi
        xs = rng.uniform(0, L, size=(20000, 1))
        ni = compute_n(xs, dxi, Ngi, n0, L, 20000, False, "CIC")
        G = generate_grad(L, Ngi)
        Lap = generate_laplacian(L, Ngi)
        g2[f"n_{Ngi}"] = ni
        for gam in (5.0, 0.3):
            phi = Gaussian_Elimination_Periodic(Lap, ni - n0, gam)
            g2[f"phi_{Ngi}_g{gam}"] = phi
            g2[f"E_{Ngi}_g{gam}"] = ((-1) * G @ phi.reshape(-1, 1))[:, 0]
    save("g2_solve", L=L, n0=n0, **g2)

    # ---- G3: compute_E with / without E_ext ----------------------------------
    Ng, N = 128, 4000
    dx = L / Ng
    x = rng.uniform(-0.5 * L, 1.5 * L, size=(N, 1))
    act = E_field(L, Ng, 3)
    cc, cs = rng.uniform(-1.25, 1.25, 3), rng.uniform(-1.25, 1.25, 3)
    E_ext = act.compute_E(cc, cs)
    E0, phi0, Em0, pm0 = compute_E(x.copy(), dx, Ng, n0, L, N, None, None, True, "CIC", None)
    E1, Em1 = compute_E(x.copy(), dx, Ng, n0, L, N, None, None, False, "CIC", E_ext)
    Et, Emt = compute_E(x.copy(), dx, Ng, n0, L, N, None, None, False, "TSC", E_ext)
    save("g3_compute_E", L=L, Ng=Ng, n0=n0, x=x, E_ext=E_ext, E=E0, phi=phi0, E_mesh=Em0, phi_mesh=pm0,
         E_with_ext=E1, E_mesh_with_ext=Em1, tsc_E_with_ext=Et, tsc_E_mesh_with_ext=Emt,
         PE=compute_electric_energy(x.copy(), dx, N, Ng, n0, L, "CIC"),
         H=compute_hamiltonian(x.copy(), 0.5 * x.copy(), dx, N, Ng, n0, L, "CIC"))

    # ---- G4/G5/G6: steps and trajectories -------------------------------------
    def trajectory(tag, dist, N, Ng, dt, K_marks, n_mode, A, interpol="CIC", with_ext=False, max_mode=3):
        sim = PIC(N=N, N_mesh=Ng, n0=1.0, L=L, dt=dt, tmin=0.0, tmax=50.0, gamma=5.0, A=A, n_mode=n_mode,
                  interpol=interpol, init_dist=dist)
        out = dict(L=L, Ng=Ng, N=N, n0=1.0, dt_in=dt, dt=sim.dt, A=A, n_mode=n_mode, gamma=5.0,
                   x0_raw=sim.init_dist.x_init.copy(), v0_raw=sim.init_dist.v_init.copy(),
                   x_init=sim.x.copy(), v_init=sim.v.copy(), n_init=sim.n.copy(),
                   E_mesh_init=sim.E_mesh.copy(), E_init=sim.E.copy())
        rew = Reward(sim.init_dist.get_init_state(), Ng, L, -25.0, 25.0, 1.0, 1.0, 1.0)
        actu = E_field(L, Ng, max_mode)
        arng = np.random.default_rng(99)
        K = max(K_marks)
        H, PE, KE, PEr, R, acts = [], [], [], [], [], []
        H.append(sim.get_energy()); PE.append(sim.get_electric_energy()); KE.append(0.5 * np.sum(sim.v * sim.v))
        for k in range(1, K + 1):
            state = sim.get_state()
            if with_ext:
                a = arng.uniform(-1.25, 1.25, 2 * max_mode)
                actu.update_E(a[:max_mode], a[max_mode:])
                E_ext = actu.compute_E()
            else:
                a = np.zeros(2 * max_mode)
                E_ext = None
            acts.append(a)
            R.append(rew.compute_reward(state, a))          # pre-step state, as ddpg.py:455
            PEr.append(rew.compute_electric_energy(state))
            sim.update_state(E_ext)
            H.append(sim.get_energy()); PE.append(sim.get_electric_energy()); KE.append(0.5 * np.sum(sim.v * sim.v))
            if k == 1:
                out.update(x_1=sim.x.copy(), v_1=sim.v.copy(), n_1=sim.n.copy(), E_mesh_1=sim.E_mesh.copy(),
                           phi_mesh_1=sim.phi_mesh.copy(), E_1=sim.E.copy(), indx_l_1=sim.indx_l.copy(),
                           indx_r_1=sim.indx_r.copy(), weight_l_1=sim.weight_l.copy(), weight_r_1=sim.weight_r.copy())
            if k in K_marks:
                out[f"x_{k}"] = sim.x.copy(); out[f"v_{k}"] = sim.v.copy()
                out[f"E_mesh_{k}"] = sim.E_mesh.copy(); out[f"n_{k}"] = sim.n.copy()
        out.update(H=np.array(H), PE=np.array(PE), KE=np.array(KE), PE_reward=np.array(PEr), reward=np.array(R),
                   actions=np.array(acts), K_marks=np.array(sorted(K_marks)))
        save(tag, **out)

    np.random.seed(42)
    trajectory("g5_bump_on_tail_N10000_Ng128", BumpOnTail(a=0.2, v0=3.0, sigma=1.0, n_samples=10000, L=L),
               10000, 128, 0.1, (10, 100, 500), 2, 0.1)
    np.random.seed(43)
    trajectory("g5_two_stream_N5000_Ng250", TwoStream(v0=3.0, sigma=1.0, n_samples=5000, L=L),
               5000, 250, 0.1, (10, 100, 500), 2, 0.1)
    np.random.seed(44)
    trajectory("g4_bump_on_tail_ext_N4000_Ng256", BumpOnTail(a=0.2, v0=3.0, sigma=1.0, n_samples=4000, L=L),
               4000, 256, 0.1, (1, 20), 2, 0.1, with_ext=True)
    np.random.seed(45)
    trajectory("g4_two_stream_ext_N3000_Ng200", TwoStream(v0=3.0, sigma=1.0, n_samples=3000, L=L),
               3000, 200, 0.05, (1, 20), 3, 0.05, with_ext=True, max_mode=5)
    np.random.seed(46)
    trajectory("g4_tsc_bump_on_tail_ext_N3000_Ng128", BumpOnTail(a=0.2, v0=3.0, sigma=1.0, n_samples=3000, L=L),
               3000, 128, 0.1, (1, 20), 2, 0.1, interpol="TSC", with_ext=True)

    # ---- G7: CFL clamp --------------------------------------------------------
    cfl = []
    for (N, Lc, dt) in ((5000, 50.0, 0.1), (20000, 50.0, 0.1), (20000, 50.0, 0.05), (4000, 10.0, 0.2), (1000, 50.0, 1.0)):
        np.random.seed(5)
        sim = PIC(N=N, N_mesh=64, n0=1.0, L=Lc, dt=dt, init_dist=TwoStream(3.0, 1.0, N, Lc), A=0.1, n_mode=2)
        cfl.append((N, Lc, dt, sim.dt))
    save("g7_cfl", table=np.array(cfl))

    # ---- G8: actuator ---------------------------------------------------------
    g8 = {}
    for (Ngi, mm) in ((128, 3), (250, 5), (256, 1)):
        a = E_field(L, Ngi, mm)
        cc, cs = rng.uniform(-1.25, 1.25, mm), rng.uniform(-1.25, 1.25, mm)
        g8[f"cc_{Ngi}_{mm}"] = cc; g8[f"cs_{Ngi}_{mm}"] = cs
        g8[f"E_{Ngi}_{mm}"] = a.compute_E(cc, cs)
    save("g8_actuator", L=L, **g8)

    # ---- G9: spectrum ---------------------------------------------------------
    d = np.load(os.path.join(OUT, "g5_two_stream_N5000_Ng250.npz"))
    snap = np.concatenate([np.concatenate([d["x_10"], d["v_10"]], 0), np.concatenate([d["x_100"], d["v_100"]], 0)], 1)
    ks, Ek = compute_E_k_spectrum(1.0, L, L / 250, 250, snap.copy(), False)
    save("g9_spectrum", ks=ks[:8], Ek=Ek[:8, :])

    # ---- G11: phase-space histogram + KL diagnostic (objective.py:8-18, reward.py:43-46) ----
    from src.control.objective import estimate_f
    st0 = np.concatenate([d["x0_raw"].reshape(-1, 1), d["v0_raw"].reshape(-1, 1)], 0)
    st1 = np.concatenate([d["x_100"], d["v_100"]], 0)
    # edge cases for the binning: values on interior edges, on both outer edges and outside the range
    st2 = st1.copy()
    Np = st2.shape[0] // 2
    st2[:6, 0] = [0.0, L, 25.0, L / 64 * 3, np.nextafter(L, 0.0), 12.5]
    st2[Np:Np + 6, 0] = [-25.0, 25.0, 0.0, 50.0 / 64 * 5 - 25.0, 30.0, -26.0]
    rw = Reward(st0, 64, L, -25.0, 25.0, 1.0, 1.0, 1.0)
    save("g11_phase_hist", L=L, nbins=64, vmin=-25.0, vmax=25.0, n0=1.0, st0=st0, st1=st1, st2=st2,
         f0=estimate_f(st0, 64, L, -25.0, 25.0, 1.0), f1=estimate_f(st1, 64, L, -25.0, 25.0, 1.0),
         f2=estimate_f(st2, 64, L, -25.0, 25.0, 1.0), kl1=rw.compute_kl_divergence(st1),
         kl2=rw.compute_kl_divergence(st2), kl0=rw.compute_kl_divergence(st0))

    # ---- samplers (pins the product's own dist.py to the reference's RNG use) --
    np.random.seed(7)
    ts = TwoStream(v0=3.0, sigma=1.0, n_samples=2001, L=L)
    ts_x1, ts_v1 = ts.get_sample()
    ts.reinit()
    ts_x2, ts_v2 = ts.get_sample()
    np.random.seed(7)
    bt = BumpOnTail(a=0.2, v0=3.0, sigma=1.0, n_samples=2001, L=L)
    bt_x1, bt_v1 = bt.get_sample()
    bt.reinit()
    bt_x2, bt_v2 = bt.get_sample()
    save("g10_samplers", seed=7, n=2001, L=L, ts_x1=ts_x1, ts_v1=ts_v1, ts_x2=ts_x2, ts_v2=ts_v2,
         bt_x1=bt_x1, bt_v1=bt_v1, bt_x2=bt_x2, bt_v2=bt_v2, bt_high_indx=bt.high_indx,
         ts_init_state=ts.get_init_state(), bt_init_state=bt.get_init_state())


def composed_loops(only=()):
    """G12-G15 (round 4): the reference's own LOOPS around the step, run as it writes them -- the closed feedback loop of
    run_feedback.py:130-168, PIC.simulate (src/env/pic.py:175-223), the behaviour-cloning rollout of
    src/control/rl/ddpg.py:364-381 with its hard-coded spectrum arguments (n0 = 1, L = 50, Ng = 250), and
    PIC.update_state_w_input_func (pic.py:148-163) with a pure function of the sub-stage state."""
    from src.env.pic import PIC
    from src.env.dist import BumpOnTail, TwoStream
    from src.control.actuator import E_field
    from src.control.rl.reward import Reward
    from src.interpret.spectrum import compute_E_k_spectrum
    L = 50.0

    def want(tag):
        return not only or tag in only

    # ---- G12: run_feedback.py:130-168, its defaults (two-stream, N = 5000, Ng = 250, dt = 0.1) with max_mode = 5 ----------
    if want("g12"):
        N, Ng, mm, K = 5000, 250, 5, 50
        np.random.seed(47)
        sim = PIC(N=N, N_mesh=Ng, n0=1.0, L=L, dt=0.1, tmin=0.0, tmax=50.0, gamma=5.0, A=0.1, n_mode=2, interpol="CIC",
                  init_dist=TwoStream(v0=3.0, sigma=1.0, n_samples=N, L=L))
        out = dict(L=L, Ng=Ng, N=N, n0=1.0, dt=sim.dt, max_mode=mm, x_init=sim.x.copy(), v_init=sim.v.copy(),
                   init_state=sim.init_dist.get_init_state())
        actuator = E_field(L, Ng, mm)
        reward = Reward(sim.init_dist.get_init_state(), Ng, L, -25.0, 25.0, 1.0, 1.0, 1.0)
        cc, cs, Hs, PEs, rs, kl, ee, ie, fields = [], [], [], [], [], [], [], [], []
        for k in range(1, K + 1):
            _, Eks = compute_E_k_spectrum(1.0, L, L / Ng, Ng, sim.get_state(), False)
            Eks = Eks[1:mm + 1, :]
            actuator.update_E((-1) * np.real(Eks), (+1) * np.imag(Eks))
            coeffs = np.concatenate([actuator.coeff_cos.ravel(), actuator.coeff_sin.ravel()])
            E_external = actuator.compute_E()
            sim.update_state(E_external)
            cc.append(actuator.coeff_cos.ravel().copy()); cs.append(actuator.coeff_sin.ravel().copy())
            fields.append(E_external[:, 0].copy())
            Hs.append(sim.get_energy()); PEs.append(sim.get_electric_energy())
            kl.append(reward.compute_kl_divergence(sim.get_state()))
            ee.append(reward.compute_electric_energy(sim.get_state()))
            ie.append(reward.compute_input_energy(coeffs))
            rs.append(reward.compute_reward(sim.get_state(), E_external))     # the mesh field as "action" (run_feedback.py:160)
            if k in (1, 10, 50):
                out[f"x_{k}"] = sim.x.copy(); out[f"v_{k}"] = sim.v.copy(); out[f"E_mesh_{k}"] = sim.E_mesh.copy()
                out[f"n_{k}"] = sim.n.copy()
        out.update(coeff_cos=np.array(cc), coeff_sin=np.array(cs), E_external=np.array(fields), H=np.array(Hs), PE=np.array(PEs),
                   reward=np.array(rs), cost_kl=np.array(kl), cost_ee=np.array(ee), cost_ie=np.array(ie))
        save("g12_feedback_two_stream_N5000_Ng250", **out)

    # ---- G13: PIC.simulate with and without a field trajectory --------------------------------------------------------------
    if want("g13"):
        N, Ng = 3000, 128
        np.random.seed(48)
        sim = PIC(N=N, N_mesh=Ng, n0=1.0, L=L, dt=0.1, tmin=0.0, tmax=1.2, gamma=5.0, A=0.1, n_mode=2, interpol="CIC",
                  init_dist=TwoStream(v0=3.0, sigma=1.0, n_samples=N, L=L))
        Nt = int(np.ceil((sim.tmax - sim.tmin) / sim.dt))
        arng = np.random.default_rng(1348)
        act = E_field(L, Ng, 3)
        traj = [act.compute_E(arng.uniform(-1.25, 1.25, 3), arng.uniform(-1.25, 1.25, 3)) for _ in range(Nt)]
        out = dict(L=L, Ng=Ng, N=N, n0=1.0, dt=sim.dt, tmin=0.0, tmax=1.2, Nt=Nt, x_init=sim.x.copy(), v_init=sim.v.copy(),
                   E_external_traj=np.array([t[:, 0] for t in traj]))
        snapshot, E, PE = sim.simulate(traj)
        out.update(snapshot=snapshot, E=E, PE=PE, E_mesh_final=sim.E_mesh.copy())
        np.random.seed(49)
        N2, Ng2 = 2000, 64
        sim = PIC(N=N2, N_mesh=Ng2, n0=1.0, L=L, dt=0.1, tmin=0.0, tmax=0.8, gamma=5.0, A=0.1, n_mode=2, interpol="TSC",
                  init_dist=BumpOnTail(a=0.2, v0=3.0, sigma=1.0, n_samples=N2, L=L))
        out.update(free_N=N2, free_Ng=Ng2, free_tmax=0.8, free_x_init=sim.x.copy(), free_v_init=sim.v.copy())
        snapshot, E, PE = sim.simulate(None)
        out.update(free_snapshot=snapshot, free_E=E, free_PE=PE)
        save("g13_simulate", **out)

    # ---- G14: behaviour-cloning rollout, ddpg.py:364-381 (the spectrum call is hard-coded to n0 = 1, L = 50, Ng = 250) ------
    if want("g14"):
        out = dict(L=L)
        for tag, N, Ng, seed in (("a", 4000, 250, 50), ("b", 3000, 200, 51)):          # b: the environment's own mesh is NOT 250
            np.random.seed(seed)
            env = PIC(N=N, N_mesh=Ng, n0=1.0, L=L, dt=0.1, tmin=0.0, tmax=50.0, gamma=5.0, A=0.1, n_mode=2, interpol="CIC",
                      init_dist=BumpOnTail(a=0.2, v0=3.0, sigma=1.0, n_samples=N, L=L))
            actuator = E_field(env.L, env.N_mesh, 5)
            reward_cls = Reward(env.init_dist.get_init_state(), env.N_mesh, env.L, -25.0, 25.0, env.n0, 1.0, 0.5)
            max_mode, K = 5, 25
            out.update({f"{tag}_N": N, f"{tag}_Ng": Ng, f"{tag}_dt": env.dt, f"{tag}_x_init": env.x.copy(),
                        f"{tag}_v_init": env.v.copy(), f"{tag}_init_state": env.init_dist.get_init_state()})
            actions, rewards = [], []
            for idx_t in range(K):
                state = env.get_state()
                _, Eks = compute_E_k_spectrum(1.0, 50.0, 50.0 / 250, 250, state, False)
                Eks = Eks[1:max_mode + 1, :]
                actuator.update_E((-1) * np.real(Eks), (+1) * np.imag(Eks))
                action = np.concatenate([actuator.coeff_cos.ravel(), actuator.coeff_sin.ravel()])
                env.update_state(E_external=actuator.compute_E())
                actions.append(action)
                rewards.append(reward_cls.compute_reward(state, action))      # on the PRE-step state
            out.update({f"{tag}_actions": np.array(actions), f"{tag}_reward": np.array(rewards), f"{tag}_x_final": env.x.copy(),
                        f"{tag}_v_final": env.v.copy(), f"{tag}_E_mesh_final": env.E_mesh.copy(), f"{tag}_steps": K})
        save("g14_bc_rollout", **out)

    # ---- G15: PIC.update_state_w_input_func (pic.py:148-163) with a PURE input function of the sub-stage state -----------------
    if want("g15"):
        N, Ng, K = 3000, 128, 5
        np.random.seed(52)
        sim = PIC(N=N, N_mesh=Ng, n0=1.0, L=L, dt=0.1, tmin=0.0, tmax=50.0, gamma=5.0, A=0.1, n_mode=2, interpol="CIC",
                  init_dist=TwoStream(v0=3.0, sigma=1.0, n_samples=N, L=L))
        mesh = np.linspace(0, L, Ng).reshape(-1, 1)
        calls = []

        def input_func(eta):
            calls.append(eta.copy())                 # as handed over: before compute_E wraps eta[:N] in place (util.py:51)
            a = np.mean(np.cos(2 * np.pi * eta[:N] / L))
            b = np.mean(eta[N:] ** 2)
            return 0.3 * a * np.sin(2 * np.pi * mesh / L) + 0.02 * b * np.cos(4 * np.pi * mesh / L)

        out = dict(L=L, Ng=Ng, N=N, n0=1.0, dt=sim.dt, steps=K, x_init=sim.x.copy(), v_init=sim.v.copy())
        xs, vs, Es, Hs, ncalls = [], [], [], [], []
        for k in range(K):
            calls.clear()
            sim.update_state_w_input_func(input_func)
            ncalls.append(len(calls))
            if k == 0:      # the three calls whose force the integrator uses: [q1; p0], [q2; p1], [q3; p2] (calls 1, 3, 5 of 7)
                out["useful_calls_step1"] = np.stack([calls[i][:, 0] for i in (1, 3, 5)])
            xs.append(sim.x[:, 0].copy()); vs.append(sim.v[:, 0].copy()); Es.append(sim.E_mesh[:, 0].copy()); Hs.append(sim.get_energy())
        out.update(x=np.array(xs), v=np.array(vs), E_mesh=np.array(Es), H=np.array(Hs), calls_per_step=np.array(ncalls))
        save("g15_input_func", **out)

    # ---- G16: PIC.compute_state_gradient (pic.py:125-129) on arbitrary states, and what reinit() leaves (pic.py:84-91) ---------
    if want("g16"):
        N, Ng = 2500, 96
        np.random.seed(53)
        sim = PIC(N=N, N_mesh=Ng, n0=1.0, L=L, dt=0.1, tmin=0.0, tmax=50.0, gamma=5.0, A=0.1, n_mode=2, interpol="CIC",
                  init_dist=BumpOnTail(a=0.2, v0=3.0, sigma=1.0, n_samples=N, L=L))
        grng = np.random.default_rng(1653)
        eta = np.concatenate([grng.uniform(-0.7 * L, 1.9 * L, (N, 1)), grng.normal(0, 2, (N, 1))], axis=0)   # positions outside [0, L) too
        ext = E_field(L, Ng, 4).compute_E(grng.uniform(-1, 1, 4), grng.uniform(-1, 1, 4))
        out = dict(L=L, Ng=Ng, N=N, n0=1.0, eta=eta.copy(), E_ext=ext)
        e1 = eta.copy(); g1 = sim.compute_state_gradient(e1, None)
        e2 = eta.copy(); g2 = sim.compute_state_gradient(e2, ext)
        out.update(grad_free=g1, grad_ext=g2, eta_after=e1)       # compute_E wraps eta[:N] in place (util.py:51)
        sim.update_state(None)
        out.update(x_before=sim.x.copy(), E_mesh_before=sim.E_mesh.copy())
        sim.reinit()                                               # a fresh sample from the global RNG, fields set to None
        out.update(x_reinit=sim.x.copy(), v_reinit=sim.v.copy(), n_reinit=sim.n.copy(),
                   fields_none=np.array([sim.E is None, sim.E_mesh is None, sim.phi_mesh is None]))
        sim.update_state(None)
        out.update(x_after=sim.x.copy(), v_after=sim.v.copy(), E_mesh_after=sim.E_mesh.copy(), H_after=sim.get_energy())
        save("g16_gradient_reinit", **out)


if __name__ == "__main__":
    # `make_golden.py g12 ... g16` regenerates only the named fixtures of round 4; no argument = everything
    sel = tuple(a for a in sys.argv[1:] if a in ("g12", "g13", "g14", "g15", "g16"))
    if sel:
        _import_reference()
        composed_loops(sel)
    else:
        main()
        composed_loops()
