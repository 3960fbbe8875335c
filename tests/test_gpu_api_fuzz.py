"""Randomised sequences of ABI calls against a model built on the NumPy oracle.

The handle keeps hidden state between calls -- the cached deposit of the next step's first drift, the ring of
accumulator rows with its clean / retired bookkeeping, an open staged step, which schedule steps the particles -- and
every entry point may be called in any order the header allows.  Each sequence mixes steps (with and without an
external field, one or several per call), controlled rollouts (one action held, a new action or a new mesh field
every step in one call, the feedback law on the device, a Gym-style step-and-observe call), staged steps, energy histories, resets, particle loads followed
by refresh / invalidate / nothing, probes in the middle of everything, and checks particles, fields and energies
against the oracle (src/env/pic.py:131-146 restated) after every state-changing call.  Mesh sizes are ones for which
the reference's own periodic solve is regular at L = 50 (it is singular e.g. for Ng = 8, 64, 100: DESIGN.md 2); a one-off
run of 80 further seeds over ten shapes passed as well (profiles/experiments_r2.md)."""
import numpy as np
import pytest

from conftest import circ_err, rel_err

pytestmark = pytest.mark.gpu

L = 50.0


class Model:
    """Reference state: one OraclePIC per environment, stepped with the same arguments."""

    def __init__(self, po, x0, v0, Ng, dt):
        self.po, self.Ng, self.dt = po, Ng, dt
        self.load(x0, v0)

    def load(self, x0, v0):
        self.sims = [self.po.OraclePIC(x0[e], v0[e], self.Ng, L=L, dt=self.dt, perturb=False, faithful=False)
                     for e in range(x0.shape[0])]

    def step(self, ext, n=1):
        for _ in range(n):
            for e, s in enumerate(self.sims):
                s.update_state(None if ext is None else ext[e].reshape(-1, 1))

    def energies(self):
        return (np.array([s.kinetic_energy() for s in self.sims]), np.array([s.get_electric_energy() for s in self.sims]))


def check(h, m, tag):
    x, v = h.particles()
    n, Em, phi = h.fields()
    ke, pe, per = h.energies()
    mke, mpe = m.energies()
    for e, s in enumerate(m.sims):
        assert circ_err(x[e], s.x, L) / L < 1e-11, tag
        assert rel_err(v[e], s.v) < 1e-11, tag
        assert rel_err(n[e], s.n) < 1e-11 and rel_err(Em[e], s.E_mesh) < 1e-8, tag
    assert np.allclose(ke, mke, rtol=1e-11) and np.allclose(pe, mpe, rtol=1e-7), tag
    assert h.bad_count() == 0, tag


@pytest.mark.parametrize("seed,N,Ng,bpe", [(1, 3000, 96, 0), (2, 3000, 96, 3), (3, 9000, 128, 0), (4, 700, 33, -1),
                                          (5, 700, 33, 2), (6, 5000, 250, 0)])
def test_random_call_sequences(seed, N, Ng, bpe):
    import ocplasma_amd as oc
    from oracle import pic_oracle as po

    rng = np.random.default_rng(seed)
    E_ = int(rng.integers(1, 4))
    dt = min(0.1, 2 / np.sqrt(N / L))

    def fresh():
        return rng.uniform(0, L, (E_, N)), rng.normal(0, 1.2, (E_, N))

    def field():
        return None if rng.integers(0, 3) == 0 else 0.1 * rng.normal(size=(E_, Ng))

    h = oc._abi.Handle(N, Ng, E_, L, 1.0, dt, blocks_per_env=bpe)
    M = int(rng.integers(1, 5))
    act = oc.E_field(L, Ng, M)
    h.set_actuator(act.basis_cos, act.basis_sin)
    x0, v0 = fresh()
    h.reset(x0, v0)
    m = Model(po, x0, v0, Ng, dt)
    log = [f"schedule={h.schedule()}"]
    for it in range(40):
        op = int(rng.integers(0, 14))
        if op <= 1:                                  # plain steps, one call
            ext, n = field(), int(rng.integers(1, 4))
            h.step(ext, n)
            m.step(ext, n)
            log.append(f"step x{n} ext={ext is not None}")
        elif op == 2:                                # the same step in three calls (update_state_w_input_func path)
            ext = field()
            for stage in (1, 2, 3):
                h.step_stage(stage, ext)
                if stage < 3 and rng.integers(0, 2):
                    h.eval_field(rng.uniform(0, L, (E_, N)))      # a probe between stages must not disturb the step
            m.step(ext, 1)
            log.append("staged step")
        elif op == 3:                                # energy history
            ext, n = field(), int(rng.integers(1, 4))
            ke, pe, per = h.step_history(ext, n)
            for k in range(n):
                m.step(ext, 1)
                mke, mpe = m.energies()
                assert np.allclose(ke[k], mke, rtol=1e-11) and np.allclose(pe[k], mpe, rtol=1e-7), log
            log.append(f"history x{n}")
        elif op == 4:                                # reset with new particles
            x0, v0 = fresh()
            h.reset(x0, v0)
            m.load(x0, v0)
            log.append("reset")
        elif op == 5:                                # load particles, then refresh / invalidate / nothing before the next step
            x0, v0 = fresh()
            h.set_particles(x0, v0)
            m.load(x0, v0)
            how = int(rng.integers(0, 3))
            if how == 0:
                h.refresh()
            elif how == 1:
                h.invalidate()
            if how != 0:                             # fields are only defined after a refresh or a step
                h.step(None, 1)
                m.step(None, 1)
            log.append(f"set_particles how={how}")
        elif op == 6:                                # abandon a staged step half way
            h.step_stage(1, field())
            x0, v0 = fresh()
            h.reset(x0, v0)
            m.load(x0, v0)
            log.append("abandoned stage + reset")
        elif op == 7:                                # probes: compute_E on arbitrary positions vs the oracle's deposit + solve
            xp = rng.uniform(-0.5 * L, 1.5 * L, (E_, N))
            n, Em, _ = h.eval_field(xp)
            for e in range(E_):
                u = xp[e].reshape(-1, 1).copy()
                _, Eo = po.field_at_particles(u, L / Ng, Ng, 1.0, L, N)
                assert rel_err(Em[e], Eo) < 1e-8, log
            log.append("probe")
            continue
        elif op == 9:                                # one action held for n steps (E_field.compute_E on the device)
            a, n = rng.uniform(-1.25, 1.25, (E_, 2 * M)), int(rng.integers(1, 3))
            h.step_actions(a, n)
            m.step(act.compute_E_batched(a), n)
            log.append(f"actions x{n}")
        elif op == 10:                               # a new action every step, one call, with the energy record
            n = int(rng.integers(1, 4))
            a = rng.uniform(-1.25, 1.25, (n, E_, 2 * M))
            rec = h.step_actions_traj(a, history=bool(rng.integers(0, 2)))
            for k in range(n):
                m.step(act.compute_E_batched(a[k]), 1)
                if rec is not None:
                    mke, mpe = m.energies()
                    assert np.allclose(rec[0][k], mke, rtol=1e-11) and np.allclose(rec[1][k], mpe, rtol=1e-7), log
            log.append(f"actions_traj x{n}")
        elif op == 11:                               # a new mesh field every step, one call
            n = int(rng.integers(1, 4))
            f = 0.1 * rng.normal(size=(n, E_, Ng))
            h.step_ext_traj(f)
            for k in range(n):
                m.step(f[k], 1)
            log.append(f"ext_traj x{n}")
        elif op == 12:                               # the feedback law on the device against the same law on the oracle's field
            n = int(rng.integers(1, 4))
            rec = h.step_feedback(n, actions=True)
            for k in range(n):
                Ek = np.stack([(np.fft.fft(s_.E_mesh[:, 0]) / Ng * 2.0)[1:M + 1] for s_ in m.sims])
                a = np.concatenate([-Ek.real, Ek.imag], axis=1)
                assert np.allclose(rec["actions"][k], a, rtol=1e-7, atol=1e-10), log
                m.step(act.compute_E_batched(a), 1)
            log.append(f"feedback x{n}")
        elif op == 13:                               # a Gym-style iteration: step + observation + energies in one call
            ext, n = field(), int(rng.integers(1, 3))
            a = rng.uniform(-1.25, 1.25, (E_, 2 * M)) if ext is None and rng.integers(0, 2) else None
            x, v, ke, pe, per = h.step_observe(ext, a, n)
            m.step(act.compute_E_batched(a) if a is not None else ext, n)
            mke, mpe = m.energies()
            assert np.allclose(ke, mke, rtol=1e-11) and np.allclose(pe, mpe, rtol=1e-7), log
            for e, s_ in enumerate(m.sims):
                assert circ_err(x[e], s_.x, L) / L < 1e-11 and rel_err(v[e], s_.v) < 1e-11, log
            log.append(f"step_observe x{n}")
        else:                                        # device-sampled reset: take the particles over into the model
            h.reset_sampled("two-stream", seed=int(rng.integers(0, 1000)))
            x0, v0 = h.particles()
            m.load(x0, v0)
            log.append("reset_sampled")
        check(h, m, log[-6:])
    h.close()
