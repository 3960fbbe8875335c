"""Initial-condition samplers with the interface of the reference's ``src/env/dist.py``.

Sampling is host-side (it is the reset boundary): the GPU
takes the drawn ``x0, v0``.  The samplers draw from NumPy's *global* RNG in the
same order as the reference (``x, v, u`` uniforms in batches of 1000,
dist.py:74-78 / 164-168), so seeding ``np.random.seed(s)`` reproduces the
reference's particles bit for bit (tests/golden/g10_samplers.npz).  Accepted
draws are kept as arrays, not Python lists, which is what makes N = 1e6 usable.
"""
import numpy as np

from .._params import ParamMixin


def _gaussian_pdf(v, vb, sigma):
    # dist.py:66-68 / 147-149
    return 1 / np.sqrt(2 * np.pi) / sigma * np.exp(-0.5 * (v - vb) ** 2 / sigma ** 2)


def _draw_until(count_reached, L, vb, sigma, batch):
    """Rejection-sample (x, v) batches until ``count_reached(total)``; returns arrays."""
    xs, vs, total = [], [], 0
    while not count_reached(total):
        x = np.random.uniform(0, L, size=batch)
        v = np.random.uniform(-10, 10, size=batch)
        u = np.random.uniform(0, 1.0, size=batch)
        keep = u < _gaussian_pdf(v, vb, sigma)
        xs.append(x[keep])
        vs.append(v[keep])
        total += int(keep.sum())
    if not xs:
        return np.empty(0), np.empty(0)
    return np.concatenate(xs), np.concatenate(vs)


class _Sampler(ParamMixin):
    def initialize(self, n_samples):
        state = self.rejection_sampling(n_samples)
        self.x_init = state[:, 0]
        self.v_init = state[:, 1]

    def reinit(self):
        self.initialize(self.n_samples)

    def get_sample(self):
        return self.x_init.copy(), self.v_init.copy()

    def get_init_state(self):
        return np.concatenate([self.x_init.copy().reshape(-1, 1), self.v_init.copy().reshape(-1, 1)], axis=0)


class TwoStream(_Sampler):
    """Two counter-streaming Maxwellians at +-v0 (reference dist.py:27-102)."""

    def __init__(self, v0=4.0, sigma=0.5, n_samples=40000, L=50):
        self.v0, self.sigma, self.L, self.n_samples = v0, sigma, L, n_samples
        self.initialize(n_samples)

    def get_proposal_prob(self, v):
        return np.exp(-abs(v))

    def get_target_prob(self, v, vb):
        return _gaussian_pdf(v, vb, self.sigma)

    def rejection_sampling(self, n_samples, batch=1000):
        half = n_samples // 2
        # the reference's first loop runs while len <= n//2 (dist.py:74), i.e. until MORE than half
        xp, vp = _draw_until(lambda c: c > half, self.L, self.v0, self.sigma, batch)
        xp, vp = xp[:half], vp[:half]
        xm, vm = _draw_until(lambda c: c >= n_samples - half, self.L, (-1) * self.v0, self.sigma, batch)
        out = np.zeros((n_samples, 2))
        out[:, 0] = np.concatenate([xp, xm])[:n_samples]
        out[:, 1] = np.concatenate([vp, vm])[:n_samples]
        return out


class BumpOnTail(_Sampler):
    """Thermal bulk N(0,1) plus a beam N(v0, sigma) carrying a/(1+a) of the particles
    (reference dist.py:104-194)."""

    def __init__(self, a=0.3, v0=4.0, sigma=0.5, n_samples=40000, L=10):
        self.a, self.v0, self.sigma, self.L, self.n_samples = a, v0, sigma, L, n_samples
        self.initialize(n_samples)

    def initialize(self, n_samples):
        super().initialize(n_samples)
        self.high_indx = self.inject_high_electron_indice()

    def get_proposal_prob(self, x, v):
        return np.exp(-abs(v))

    def get_target_prob(self, v, vb, sigma):
        return _gaussian_pdf(v, vb, sigma)

    def _n_bulk(self, n_samples):
        return int(n_samples * (1 / (1 + self.a)))

    def rejection_sampling(self, n_samples, batch=1000):
        n1 = self._n_bulk(n_samples)
        xb, vb_ = _draw_until(lambda c: c >= n1, self.L, 0.0, 1.0, batch)
        xb, vb_ = xb[:n1], vb_[:n1]
        xt, vt = _draw_until(lambda c: c >= n_samples - n1, self.L, self.v0, self.sigma, batch)
        out = np.zeros((n_samples, 2))
        out[:, 0] = np.concatenate([xb, xt])[:n_samples]
        out[:, 1] = np.concatenate([vb_, vt])[:n_samples]
        return out

    def inject_high_electron_indice(self):
        return np.arange(self._n_bulk(self.n_samples), self.n_samples)
