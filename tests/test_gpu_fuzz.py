"""Seeded randomized differential test: random problem shapes through the C ABI against the NumPy
oracle (which is pinned to the reference by tests/golden).  Complements the fixed cases of
test_gpu_parity.py with odd sizes, tiny and non-power-of-two meshes, both shape functions, several
environments per handle and external fields."""
import numpy as np
import pytest

from conftest import circ_err, rel_err

pytestmark = pytest.mark.gpu

CASES = 150


def _case(rng):
    N = int(rng.choice([1, 2, 3, 17, 255, 256, 257, 511, 513, 1000, 2049, 4999]))
    Ng = int(rng.choice([4, 5, 7, 16, 31, 64, 100, 127, 250, 300]))
    L = float(rng.choice([1.0, 10.0, 50.0, 77.7]))
    n0 = float(rng.choice([0.5, 1.0, 2.5]))
    dt = float(rng.choice([0.01, 0.05, 0.1]))
    interpol = str(rng.choice(["CIC", "TSC"]))
    envs = int(rng.integers(1, 5))
    ext = bool(rng.integers(0, 2))
    return N, Ng, L, n0, dt, interpol, envs, ext


def test_random_shapes_against_oracle():
    import ocplasma_amd as oc
    from oracle import pic_oracle as po

    rng = np.random.default_rng(20261004)
    checked = 0
    for _ in range(CASES):
        N, Ng, L, n0, dt, interpol, envs, ext = _case(rng)
        x0 = rng.uniform(-0.25 * L, 1.25 * L, (envs, N))          # includes positions outside the box
        v0 = rng.normal(0.0, 1.0, (envs, N)) * rng.choice([0.1, 1.0, 3.0])
        E_ext = rng.uniform(-0.5, 0.5, (envs, Ng)) if ext else None
        # alternate between the two schedules: resident (one workgroup per environment) and streaming sweeps with a
        # random number of workgroups per environment
        bpe = 0 if rng.integers(0, 2) else int(rng.integers(1, 6))
        env = oc.BatchedPIC(envs, N, Ng, n0=n0, L=L, dt=dt, interpol=interpol, blocks_per_env=bpe)
        env.reset(x0, v0)
        env.step(E_ext, nsteps=2)
        x, v = env.particles()
        n, Em, phi = env.fields()
        ke, pe, per = env.energies()
        assert env.bad_count() == 0
        tag = (N, Ng, L, n0, dt, interpol, envs, ext, bpe, env._h.schedule())
        for e in range(envs):
            try:
                with np.errstate(all="ignore"):
                    ref = po.OraclePIC(x0[e], v0[e], Ng, n0=n0, L=L, dt=dt, interpol=interpol, perturb=False,
                                       faithful=False)
                    assert ref.dt == env.dt, tag
                    for _k in range(2):
                        ref.update_state(None if E_ext is None else E_ext[e].reshape(-1, 1))
                finite = np.isfinite(ref.E_mesh).all() and np.isfinite(ref.x).all()
            except ValueError:        # NaN positions reach np.bincount, as they would in the reference
                finite = False
            if not finite:
                # the reference's Sherman-Morrison solve is singular for this (L, Ng); the device
                # solver has no such failure mode, its own invariants are checked instead
                assert np.isfinite(Em[e]).all() and abs(n[e].sum() * (L / Ng) - n0 * L) < 1e-9 * n0 * L, tag
                continue
            scale = max(1.0, float(np.max(np.abs(ref.v))))
            assert circ_err(x[e], ref.x, L) / L < 1e-11, tag
            assert np.max(np.abs(v[e] - ref.v[:, 0])) / scale < 1e-11, tag
            assert rel_err(n[e], ref.n) < 1e-11, tag
            if np.max(np.abs(ref.E_mesh)) > 1e-9:
                assert rel_err(Em[e], ref.E_mesh) < 1e-8, tag
            assert abs(ke[e] - ref.kinetic_energy()) <= 1e-11 * max(1.0, ref.kinetic_energy()), tag
            checked += 1
        env.close()
    assert checked >= CASES          # most environments have a finite reference to compare with


def test_random_shapes_function_level_mirrors():
    """The module-function drop-ins (env.util / env.interpolate / env.solve) on random shapes against the oracle:
    indices and weights bit for bit, densities / fields to rounding."""
    import ocplasma_amd  # noqa: F401
    from ocplasma_amd.env import interpolate, solve, util
    from oracle import pic_oracle as po

    rng = np.random.default_rng(77)
    solved = 0
    for _ in range(40):
        N, Ng, L, n0, _dt, interpol, _envs, ext = _case(rng)
        dx = L / Ng
        x = rng.uniform(-0.25 * L, 1.25 * L, (N, 1))
        tag = (N, Ng, L, n0, interpol, ext)
        ours = (interpolate.CIC if interpol == "CIC" else interpolate.TSC)(x, n0, L, N, Ng, dx)
        theirs = (po.cic if interpol == "CIC" else po.tsc)(x, n0, L, N, Ng, dx)
        assert len(ours) == len(theirs), tag
        assert rel_err(ours[0], theirs[0]) < 1e-12 or np.max(np.abs(ours[0] - theirs[0])) < 1e-12, tag
        k = (len(ours) - 1) // 2
        for a, b in zip(ours[1:1 + k], theirs[1:1 + k]):
            # the oracle keeps floor(x/dx) unfolded where x/dx rounds up to Ng (interpolate.py:8); the device folds it
            assert np.array_equal(a % Ng, b % Ng), tag
        for a, b in zip(ours[1 + k:], theirs[1 + k:]):
            assert np.max(np.abs(a - b)) < 1e-15 if interpol == "TSC" else np.array_equal(a, b), tag
        E_ext = rng.uniform(-0.5, 0.5, (Ng, 1)) if ext else None
        u = np.concatenate([x, rng.normal(size=(N, 1))])
        E, phi, E_mesh, phi_mesh = util.compute_E(u.copy(), dx, Ng, n0, L, N, None, None, True, interpol, E_ext)
        with np.errstate(all="ignore"):
            Eo, Emo = po.field_at_particles(u.copy(), dx, Ng, n0, L, N, None, None, interpol, E_ext)
        if not np.isfinite(Emo).all():
            continue                          # singular Sherman-Morrison in the reference algorithm
        if np.max(np.abs(Emo)) > 1e-9:
            assert rel_err(E_mesh, Emo) < 1e-8, tag
            assert np.max(np.abs(E - Eo)) < 1e-8 * np.max(np.abs(Emo)), tag     # a lone particle feels ~0 (self-force)
        assert abs(phi_mesh.mean()) < 1e-9 * max(1.0, np.abs(phi_mesh).max()), tag
        # the returned potential solves the discrete Poisson problem for the returned density
        n = util.compute_n(u.copy(), dx, Ng, n0, L, N, False, interpol)
        lap = (np.roll(phi_mesh, -1, 0) - 2 * phi_mesh + np.roll(phi_mesh, 1, 0)) / dx ** 2
        assert np.max(np.abs(lap[:, 0] - (n - n0))) < 1e-8 * max(1.0, np.abs(n).max()), tag
        phi2 = solve.Gaussian_Elimination_Periodic(util.generate_laplacian(L, Ng), n - n0)
        assert np.max(np.abs(phi2 - phi_mesh[:, 0])) < 1e-10 * max(1.0, np.abs(phi2).max()), tag
        solved += 1
    assert solved >= 25
