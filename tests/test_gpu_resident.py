"""The resident schedule (csrc/pic_resident.h): environments whose particles fit one workgroup are stepped inside a
single launch per pic_step call.  It must be indistinguishable from the streaming sweeps: particles, density, field
and potential bit for bit (every deposit is the same integer sum), energies to rounding (KE is a float64 sum whose
order follows the launch geometry) -- and, like them, agree with the NumPy oracle and the reference's golden
trajectory (src/env/pic.py:131-146)."""
import numpy as np
import pytest

from conftest import circ_err, load_golden, rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def oc():
    import ocplasma_amd
    return ocplasma_amd


@pytest.fixture(scope="module")
def po():
    from oracle import pic_oracle
    return pic_oracle


CASES = [  # N, Ng, envs, dtype, position_dtype, interpol
    (5000, 250, 7, "float64", None, "CIC"),          # the reference's default shape (run_wo_oc.py:33-34)
    (5000, 250, 3, "float64", None, "TSC"),
    (1, 8, 2, "float64", None, "CIC"),
    (513, 64, 2, "float64", None, "CIC"),
    (2048, 128, 2, "float64", None, "CIC"),          # exactly full at 4 particles per lane
    (2049, 128, 2, "float64", None, "CIC"),
    (2500, 100, 300, "float64", None, "CIC"),        # more environments than CUs: the lean kernel (two workgroups per CU)
    (5000, 250, 260, "float32", "fixed32", "CIC"),
    (8192, 1024, 2, "float64", None, "CIC"),         # the largest resident environment
    (5000, 250, 3, "float32", None, "CIC"),          # packed accumulator
    (5000, 250, 3, "float32", None, "TSC"),
    (5000, 250, 3, "float32", "fixed32", "CIC"),
    (3000, 200, 2, "float32", "fixed32", "TSC"),
]


@pytest.mark.parametrize("N,Ng,E_,dtype,pos,interpol", CASES)
def test_resident_equals_streaming(oc, N, Ng, E_, dtype, pos, interpol):
    L = 50.0
    rng = np.random.default_rng(N + Ng)
    x0 = rng.uniform(0, L, (E_, N)).astype(dtype)
    x0[x0 >= L] = 0.0
    v0 = rng.normal(0, 1.5, (E_, N)).astype(dtype)
    ext = 0.05 * rng.normal(size=(E_, Ng))
    kw = dict(L=L, dt=0.1, dtype=dtype, position_dtype=pos, interpol=interpol)
    res = oc.BatchedPIC(E_, N, Ng, blocks_per_env=-1, **kw)    # -1: resident (0 = resident where it pays)
    seq = oc.BatchedPIC(E_, N, Ng, blocks_per_env=3, **kw)     # > 0: streaming sweeps
    assert res._h.schedule() == "resident" and seq._h.schedule() == "streaming"
    for env in (res, seq):
        env.reset(x0, v0)
    for env in (res, seq):
        env.step(ext, nsteps=5)          # 5 steps inside one launch / 15 sweeps
        env.step(None, nsteps=1)
        env.step(ext)
    (xr, vr), (xs, vs) = res.particles(), seq.particles()
    assert np.array_equal(xr, xs) and np.array_equal(vr, vs)
    for a, b in zip(res.fields(), seq.fields()):
        assert np.array_equal(a, b)
    (kr, pr, rr), (ks, ps, rs) = res.energies(), seq.energies()
    assert np.array_equal(pr, ps) and np.array_equal(rr, rs) and np.allclose(kr, ks, rtol=1e-13 if dtype == "float64" else 1e-6)
    assert res.bad_count() == 0
    # per-step energy traces: one launch for all steps against step-by-step reads
    hr = res.step_history(ext, 4)
    hs = seq.step_history(ext, 4)
    assert np.array_equal(hr[1], hs[1]) and np.array_equal(hr[2], hs[2]) and np.allclose(hr[0], hs[0], rtol=1e-13 if dtype == "float64" else 1e-6)
    assert np.array_equal(res.particles()[0], seq.particles()[0])
    # the staged entry point (one launch per sub-stage) still works on a resident handle
    for stage in (1, 2, 3):
        res._h.step_stage(stage, ext)
        seq._h.step_stage(stage, ext)
    assert np.array_equal(res.particles()[0], seq.particles()[0]) and np.array_equal(res.fields()[1], seq.fields()[1])
    res.step(ext)
    seq.step(ext)
    assert np.array_equal(res.particles()[1], seq.particles()[1])
    res.close()
    seq.close()


def test_resident_matches_oracle_and_golden_trajectory(oc, po):
    """The reference's own 500-step two-stream trajectory (N = 5000, Ng = 250) through the resident schedule, in
    chunks of 100 steps per launch, and a batch of small environments with per-step actions against the oracle."""
    g = load_golden("g5_two_stream_N5000_Ng250")
    L, Ng = float(g["L"]), int(g["Ng"])
    env = oc.BatchedPIC(1, int(g["N"]), Ng, L=L, dt=float(g["dt_in"]))
    assert env._h.schedule() == "resident" and env.dt == float(g["dt"])
    env.reset(g["x_init"].reshape(1, -1), g["v_init"].reshape(1, -1))
    H = []
    for k in range(5):
        ke, pe, _ = env.step_history(None, 100)
        H.extend((ke + pe)[:, 0])
        if (k + 1) * 100 in (100, 500):
            K = (k + 1) * 100
            x, v = env.particles()
            tol = 1e-11 if K == 100 else 1e-7
            assert circ_err(x[0], g[f"x_{K}"], L) / L < tol and rel_err(v[0], g[f"v_{K}"]) < tol
            assert rel_err(env.fields()[1][0], g[f"E_mesh_{K}"]) < (5e-11 if K == 100 else 5e-8)
    assert rel_err(H, g["H"][1:]) < 1e-10
    env.close()
    # many environments, a new action every step
    E_, N, M = 40, 5000, 3
    xs, vs = zip(*[po.synthetic_two_stream(N, L, seed=500 + e) for e in range(E_)])
    act = oc.E_field(L, Ng, M)
    env = oc.BatchedPIC(E_, N, Ng, L=L, dt=0.1)
    env.set_actuator(act)
    env.reset(np.stack(xs), np.stack(vs))
    rng = np.random.default_rng(9)
    refs = {e: po.OraclePIC(xs[e], vs[e], Ng, L=L, dt=0.1, perturb=False, faithful=False) for e in (0, 17, 39)}
    for _ in range(6):
        a = rng.uniform(-1.25, 1.25, (E_, 2 * M))
        env.step_actions(a)
        ext = act.compute_E_batched(a)
        for e, ref in refs.items():
            ref.update_state(ext[e].reshape(-1, 1))
    x, v = env.particles()
    n, Em, _ = env.fields()
    ke, pe, _ = env.energies()
    for e, ref in refs.items():
        assert circ_err(x[e], ref.x, L) / L < 1e-12 and rel_err(v[e], ref.v) < 1e-12
        assert rel_err(n[e], ref.n) < 1e-12 and rel_err(Em[e], ref.E_mesh) < 1e-10
        assert abs(ke[e] / ref.kinetic_energy() - 1) < 1e-13 and abs(pe[e] / ref.get_electric_energy() - 1) < 1e-9
    env.close()


def test_resident_schedule_limits(oc):
    with pytest.raises(oc._abi.PicError, match="resident schedule needs"):
        oc.BatchedPIC(1, 20000, 64, blocks_per_env=-1)
    with pytest.raises(oc._abi.PicError, match="resident schedule needs"):
        oc.BatchedPIC(1, 1000, 64, blocks_per_env=-1, accum_dtype="float64")
    assert oc.BatchedPIC(1, 8192, 64, blocks_per_env=-1)._h.schedule() == "resident"
    assert oc.BatchedPIC(1, 8193, 64)._h.schedule() == "streaming"
    # automatic choice: small environments always, up to 8192 particles when there are many of them
    assert oc.BatchedPIC(1, 5000, 250)._h.schedule() == "resident"
    assert oc.BatchedPIC(1, 8000, 128)._h.schedule() == "streaming"
    assert oc.BatchedPIC(64, 8000, 128)._h.schedule() == "resident"
    assert oc.BatchedPIC(1, 1000, 64, accum_dtype="float64")._h.schedule() == "streaming"


@pytest.mark.parametrize("N,Ng,bpe,dtype,pos", [(5000, 250, 0, "float64", None), (5000, 250, 2, "float64", None),
                                                (30000, 128, 0, "float64", None), (3000, 96, 0, "float32", "fixed32"),
                                                (3000, 96, 2, "float32", "fixed32")])
def test_snapshots_of_every_step_equal_stepwise_reads(oc, N, Ng, bpe, dtype, pos):
    """pic_step_snapshots (what PIC.simulate records, pic.py:175-223): particles and energies of every step, kept on the
    device and read back once -- equal to stepping one step at a time and reading after each, in both schedules."""
    E_, K, L = 3, 6, 50.0
    rng = np.random.default_rng(N)
    x0 = rng.uniform(0, L, (E_, N)).astype(dtype)
    x0[x0 >= L] = 0.0
    v0 = rng.normal(0, 1.3, (E_, N)).astype(dtype)
    ext = 0.05 * rng.normal(size=(E_, Ng))
    kw = dict(L=L, dt=0.1, dtype=dtype, position_dtype=pos, blocks_per_env=bpe)
    a, b = oc.BatchedPIC(E_, N, Ng, **kw), oc.BatchedPIC(E_, N, Ng, **kw)
    for env in (a, b):
        env.reset(x0, v0)
    xs, vs, ke, pe, per = a.simulate_snapshots(K, ext)
    assert xs.shape == vs.shape == (K, E_, N) and ke.shape == (K, E_) and xs.dtype == np.dtype(dtype)
    for k in range(K):
        b.step(ext)
        xb, vb = b.particles()
        kb, pb, rb = b.energies()
        assert np.array_equal(xs[k], xb) and np.array_equal(vs[k], vb), k
        assert np.array_equal(pe[k], pb) and np.array_equal(per[k], rb) and np.allclose(ke[k], kb, rtol=1e-13 if dtype == "float64" else 1e-6)
    # the handle carries on from the last snapshot
    a.step(ext)
    b.step(ext)
    assert np.array_equal(a.particles()[0], b.particles()[0])
    a.close()
    b.close()


def test_simulate_of_the_drop_in_uses_one_read_back(oc):
    """PIC.simulate (pic.py:175-223) on the reference's golden two-stream run: snapshot, E and PE traces of the first 60
    steps, recorded on the device, against the reference's own per-step values."""
    g = load_golden("g5_two_stream_N5000_Ng250")

    class Fixed:
        def reinit(self):
            pass

        def get_sample(self):
            return g["x0_raw"].copy(), g["v0_raw"].copy()

    sim = oc.PIC(N=int(g["N"]), N_mesh=int(g["Ng"]), n0=float(g["n0"]), L=float(g["L"]), dt=float(g["dt_in"]), tmin=0.0,
                 tmax=60 * float(g["dt"]) - 1e-9, gamma=float(g["gamma"]), A=float(g["A"]), n_mode=int(g["n_mode"]), init_dist=Fixed())
    snap, E, PE = sim.simulate(None)
    assert snap.shape == (2 * sim.N, 61) and E.shape == PE.shape == (61,)
    assert rel_err(E, g["H"][:61]) < 1e-10 and rel_err(PE, g["PE"][:61]) < 1e-10
    assert np.array_equal(snap[:sim.N, 0:1], g["x_init"]) and np.array_equal(snap[sim.N:, 0:1], g["v_init"])
    assert circ_err(snap[:sim.N, 10], g["x_10"], float(g["L"])) / float(g["L"]) < 1e-12 and rel_err(snap[sim.N:, 10], g["v_10"]) < 1e-12
    assert np.array_equal(snap[:sim.N, 60:61], sim.x) and np.array_equal(snap[sim.N:, 60:61], sim.v)
    sim.close()
