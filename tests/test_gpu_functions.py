"""Function-level drop-ins of src/env/{util,interpolate,solve}.py against the golden
vectors the reference produced (tests/golden/g1-g3).  Everything runs through the C ABI (pic_compute_E,
pic_eval_field, pic_solve_poisson)."""
import numpy as np
import pytest

from conftest import load_golden, rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def oc():
    import ocplasma_amd
    return ocplasma_amd


@pytest.fixture(scope="module")
def mods():
    import ocplasma_amd  # noqa: F401
    from ocplasma_amd.env import interpolate, solve, util
    return util, interpolate, solve


def test_CIC_and_TSC_functions_match_the_reference(mods):
    util, interp, _ = mods
    g = load_golden("g1_deposit")
    L, Ng, n0 = float(g["L"]), int(g["Ng"]), float(g["n0"])
    dx = L / Ng
    x = g["x"].copy()
    n, jl, jr, wl, wr = interp.CIC(x, n0, L, x.shape[0], Ng, dx)
    assert np.array_equal(x, g["x"])                                   # CIC does not modify its input
    assert jl.shape == (x.shape[0], 1) and jl.dtype == np.int64 and wl.shape == (x.shape[0], 1)
    assert rel_err(n, g["n"]) < 1e-13
    assert np.array_equal(jl, g["jl"]) and np.array_equal(jr, g["jr"])
    assert np.array_equal(wl, g["wl"]) and np.array_equal(wr, g["wr"])  # bit for bit
    # edge inputs: 0, just below L, negative, beyond L
    xd = g["xin"].copy()
    n, jl, jr, wl, wr = interp.CIC(xd, n0, L, xd.shape[0], Ng, dx)
    assert rel_err(n, g["n_d"]) < 1e-13
    assert np.array_equal(jl, g["jl_d"]) and np.array_equal(jr, g["jr_d"])
    assert np.array_equal(wl, g["wl_d"]) and np.array_equal(wr, g["wr_d"])
    n, jl, jm, jr, wl, wm, wr = interp.TSC(x, n0, L, x.shape[0], Ng, dx)
    assert rel_err(n, g["tsc_n"]) < 1e-13
    assert np.array_equal(jl, g["tsc_jl"]) and np.array_equal(jm, g["tsc_jm"]) and np.array_equal(jr, g["tsc_jr"])
    for ours, ref in ((wl, g["tsc_wl"]), (wm, g["tsc_wm"]), (wr, g["tsc_wr"])):
        assert np.max(np.abs(ours - ref)) < 1e-15


def test_compute_n_and_compute_E_functions(mods):
    util, _, _ = mods
    g = load_golden("g3_compute_E")
    L, Ng, n0 = float(g["L"]), int(g["Ng"]), float(g["n0"])
    N, dx = g["x"].shape[0], L / Ng
    # reference-style call: u is the (2N, 1) state, only u[:N] is read and it is wrapped in place
    u = np.concatenate([g["x"] + L, 0.5 * g["x"]])                    # shifted by a box length on purpose
    expect_wrapped = np.mod(u[:N], L)
    E, E_mesh = util.compute_E(u, dx, Ng, n0, L, N)
    assert np.array_equal(u[:N], expect_wrapped) and np.array_equal(u[N:], 0.5 * g["x"])
    assert E.shape == (N, 1) and E_mesh.shape == (Ng, 1)
    assert rel_err(E_mesh, g["E_mesh"]) < 1e-11 and rel_err(E, g["E"]) < 1e-11   # x + L - L costs a few ulp of x
    u = np.concatenate([g["x"], 0.5 * g["x"]])
    E, phi, E_mesh, phi_mesh = util.compute_E(u, dx, Ng, n0, L, N, util.generate_grad(L, Ng),
                                              util.generate_laplacian(L, Ng), True, "CIC", g["E_ext"])
    assert rel_err(E_mesh, g["E_mesh_with_ext"]) < 1e-12 and rel_err(E, g["E_with_ext"]) < 1e-12
    gauge = g["phi_mesh"].mean()                                        # the reference's gauge is round-off
    assert rel_err(phi_mesh, g["phi_mesh"] - gauge) < 1e-10
    assert rel_err(phi, g["phi"] - gauge) < 1e-10
    assert abs(phi_mesh.mean()) < 1e-12
    E, E_mesh = util.compute_E(u, dx, Ng, n0, L, N, None, None, False, "TSC", g["E_ext"])
    assert rel_err(E_mesh, g["tsc_E_mesh_with_ext"]) < 1e-12 and rel_err(E, g["tsc_E_with_ext"]) < 1e-12

    g1 = load_golden("g1_deposit")
    L1, Ng1 = float(g1["L"]), int(g1["Ng"])
    x = g1["x"].copy()
    n = util.compute_n(x, L1 / Ng1, Ng1, float(g1["n0"]), L1, x.shape[0])
    assert n.shape == (Ng1,) and rel_err(n, g1["n"]) < 1e-13
    assert np.array_equal(x, g1["x_wrapped"])                           # single np.mod, as util.py:51 leaves it
    out = util.compute_n(g1["x"].copy(), L1 / Ng1, Ng1, float(g1["n0"]), L1, x.shape[0], True, "TSC")
    assert len(out) == 7 and np.array_equal(out[2], g1["tsc_jm"]) and np.max(np.abs(out[5] - g1["tsc_wm"])) < 1e-15
    out = util.compute_n(g1["x"].copy(), L1 / Ng1, Ng1, float(g1["n0"]), L1, x.shape[0], True)
    assert len(out) == 5 and np.array_equal(out[1], g1["jl"]) and np.array_equal(out[4], g1["wr"])


def test_energy_functions(mods):
    util, _, _ = mods
    g = load_golden("g3_compute_E")
    L, Ng, n0 = float(g["L"]), int(g["Ng"]), float(g["n0"])
    N, dx = g["x"].shape[0], L / Ng
    x, v = g["x"].copy(), 0.5 * g["x"]
    assert abs(util.compute_electric_energy(x, dx, N, Ng, n0, L) / float(g["PE"]) - 1) < 1e-12
    assert abs(util.compute_hamiltonian(x, v, dx, N, Ng, n0, L) / float(g["H"]) - 1) < 1e-13


def test_periodic_solver_function(mods):
    util, _, solve = mods
    g = load_golden("g2_solve")
    L, n0 = float(g["L"]), float(g["n0"])
    for Ng in (128, 250, 256, 1024):
        A = util.generate_laplacian(L, Ng)
        b = g[f"n_{Ng}"] - n0
        phi = solve.Gaussian_Elimination_Periodic(A, b, gamma=5.0)
        ref = g[f"phi_{Ng}_g5.0"]
        assert phi.shape == (Ng,) and abs(phi.mean()) < 1e-12
        # the reference's phi carries a round-off gauge and the conditioning of its singular solve
        assert rel_err(phi, ref - ref.mean()) < 1e-9, Ng
        assert np.max(np.abs(A @ phi - (b - b.mean()))) < 1e-9 * np.max(np.abs(b))      # it solves the system
        phi2, E = solve.solve_periodic_poisson(b, L / Ng)
        assert np.array_equal(phi2, phi)
        assert rel_err(E, g[f"E_{Ng}_g5.0"]) < 5e-12 and rel_err(E, g[f"E_{Ng}_g0.3"]) < 5e-11
    with pytest.raises(ValueError):
        solve.Gaussian_Elimination_Periodic(np.eye(16), np.zeros(16))                 # not a Laplacian
    with pytest.raises(ValueError):
        solve.Gaussian_Elimination_Periodic(util.generate_laplacian(L, 16), np.zeros(15))


def test_cfl_clamp_table_through_the_drop_in_classes(oc):
    """golden g7: PIC.initialize's CFL clamp dt = min(dt, 2 / sqrt(N / L)) (pic.py:71-73) as the reference applied it,
    through PIC and BatchedPIC (the oracle is checked against the same table in tests/test_oracle_golden.py)."""
    from conftest import load_golden

    class Uniform:
        def __init__(self, N, L):
            self.N, self.L = N, L

        def reinit(self):
            pass

        def get_sample(self):
            rng = np.random.default_rng(self.N)
            return rng.uniform(0, self.L, self.N), rng.normal(0, 1, self.N)

    for N, L, dt_in, dt_ref in load_golden("g7_cfl")["table"]:
        N = int(N)
        sim = oc.PIC(N=N, N_mesh=64, L=float(L), dt=float(dt_in), init_dist=Uniform(N, float(L)))
        assert sim.dt == dt_ref
        sim.update_state(None)                       # the handle really steps with the clamped dt
        assert sim._ensure_handle().cfg.dt == dt_ref
        sim.close()
        env = oc.BatchedPIC(2, N, 64, L=float(L), dt=float(dt_in))
        assert env.dt == dt_ref and env._h.cfg.dt == dt_ref
        env.close()


def test_dense_operator_attributes_of_the_drop_in(oc):
    """PIC.grad / PIC.laplacian (pic.py:52-53): the reference object's dense operators, for callers that read them."""
    from oracle import pic_oracle as po

    class D:
        def reinit(self):
            pass

        def get_sample(self):
            return np.linspace(0, 49, 500), np.zeros(500)

    sim = oc.PIC(N=500, N_mesh=32, L=50.0, dt=0.1, init_dist=D())
    assert np.array_equal(sim.grad, po.dense_grad(50.0, 32)) and np.array_equal(sim.laplacian, po.dense_laplacian(50.0, 32))
    assert sim.grad is sim.grad                     # built once
    # E_mesh = -grad @ phi_mesh, laplacian @ phi_mesh = n - n0 hold for the device fields (zero-mean gauge)
    sim.update_state(None)
    assert rel_err(-sim.grad @ sim.phi_mesh, sim.E_mesh) < 1e-10
    assert np.max(np.abs(sim.laplacian @ sim.phi_mesh - (sim.n.reshape(-1, 1) - sim.n0))) < 1e-9
    sim.close()


def test_integration_md_ctypes_stub_runs_verbatim(oc):
    """the reference-side ctypes binding a maintainer would add, executed as printed (only
    the library name is made absolute) and checked against the reference's own one-step output (golden g4)."""
    import os
    import re
    from conftest import ROOT, circ_err, load_golden
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    sec = text[text.index("## B."):text.index("## C.")]
    code = re.search(r"```python\n(.*?)```", sec, re.S).group(1)
    assert "class HipStepper" in code and "pic_create" in code
    code = code.replace('C.CDLL("libpicstep.so")', f'C.CDLL({oc._abi.library_path()!r})')
    ns = {}
    exec(compile(code, "INTEGRATION.md#B", "exec"), ns)
    g = load_golden("g4_bump_on_tail_ext_N4000_Ng256")

    class RefPIC:                                     # the attributes the stub reads off the reference object
        N, N_mesh, L, n0, dt, gamma, interpol = int(g["N"]), int(g["Ng"]), float(g["L"]), float(g["n0"]), float(g["dt"]), float(g["gamma"]), "CIC"

    st = ns["HipStepper"](RefPIC)
    st.reset(g["x_init"][:, 0], g["v_init"][:, 0])
    act = oc.E_field(RefPIC.L, RefPIC.N_mesh, 3)
    a = g["actions"][0]
    act.update_E(a[:3], a[3:])
    st.update_state(act.compute_E()[:, 0])
    x, v, n, E, phi = st.fetch()
    assert circ_err(x, g["x_1"], RefPIC.L) / RefPIC.L < 1e-13 and rel_err(v, g["v_1"]) < 1e-13
    assert rel_err(n, g["n_1"]) < 1e-12 and rel_err(E, g["E_mesh_1"]) < 1e-11
    H, PE, PEr = st.energies()
    assert abs(H / float(g["H"][1]) - 1) < 1e-12
    a = g["actions"][1]                               # the second golden step through the stub's one-call iteration
    act.update_E(a[:3], a[3:])
    obs, H2, PEr2 = st.step(act.compute_E()[:, 0])
    assert abs(H2 / float(g["H"][2]) - 1) < 1e-12 and abs(PEr2 / float(g["PE_reward"][2]) - 1) < 1e-10
    x, v = st.fetch()[:2]
    assert np.array_equal(obs[:RefPIC.N], x) and np.array_equal(obs[RefPIC.N:], v)


def test_c_program_drives_the_abi(oc, tmp_path):
    """examples/c_api_demo.c: a plain C process (no Python, no torch) creates a handle, resets, steps and reads back
    through include/picstep.h; its particles, fields and energies equal the ctypes binding's bit for bit."""
    import os
    import shutil
    import subprocess
    from conftest import ROOT
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc on this box")
    lib = oc._abi.library_path()
    csrc = os.path.dirname(lib)
    exe = tmp_path / "c_api_demo"
    cmd = [gcc, "-std=c99", "-O2", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "c_api_demo.c"), "-o", str(exe),
           "-L", csrc, "-lpicstep", f"-Wl,-rpath,{csrc}", "-Wl,-rpath,/opt/rocm/lib"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    for N, Ng, E_, steps in ((5000, 250, 3, 7), (30000, 128, 2, 4)):        # resident and streaming schedules
        rng = np.random.default_rng(N)
        x0, v0 = rng.uniform(0, 50.0, (E_, N)), rng.normal(0, 1.5, (E_, N))
        inp, out = tmp_path / f"in_{N}.bin", tmp_path / f"out_{N}.bin"
        ref = oc.BatchedPIC(E_, N, Ng, L=50.0, dt=0.1)
        with open(inp, "wb") as f:       # the C caller passes dt AFTER the CFL clamp (pic.py:71-73), as the header says
            f.write(np.int64(N).tobytes() + np.int32(Ng).tobytes() + np.int32(E_).tobytes() + np.float64(50.0).tobytes() + np.float64(ref.dt).tobytes())
            f.write(x0.tobytes() + v0.tobytes())
        env = dict(os.environ, LD_LIBRARY_PATH=csrc + ":/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
        r = subprocess.run([str(exe), str(inp), str(out), str(steps)], capture_output=True, text=True, timeout=120, env=env)
        assert r.returncode == 0, r.stdout + r.stderr
        assert ("resident" if N <= 5120 else "streaming") in r.stdout
        got = np.fromfile(out)
        ref.reset(x0, v0)
        ref.step(None, steps)
        x, v = ref.particles()
        n, Em, phi = ref.fields()
        ke, pe, per = ref.energies()
        want = np.concatenate([a.ravel() for a in (x, v, n, Em, phi, ke, pe, per)])
        assert got.shape == want.shape and np.array_equal(got, want)
        ref.close()
