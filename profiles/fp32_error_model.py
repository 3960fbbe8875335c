#!/usr/bin/env python3
"""Where does the float32 step lose accuracy?  CPU emulation of the HIP sweep arithmetic (csrc/pic_sweep.h) with the
precision of positions, velocities and shape weights chosen independently, run next to the all-float64 emulation
from the same float32-representable start and with the same per-step actions (BASELINE config 3 shape: two-stream,
Ng = 512, a new random E_in action every step; N reduced with --particles to keep the run short).

    python profiles/fp32_error_model.py --particles 1000000 --steps 500 --out profiles/fp32_error_model.md

variants (positions / velocities / weights):
    f32        f32 / f32 / f32         what particle_dtype = float32 does
    x64        f64 / f32 / f64         only the velocities (and the kick arithmetic) are single
    v64        f32 / f64 / f32         only the positions (and locate, weights) are single
    w64        f32 / f32 / f64         single storage, locate and weights evaluated in double from the stored floats
    u32        u32 / f32 / f32         positions as 32-bit fixed point x = u L / 2^32 (wrap = integer overflow,
                                       cell = high bits of u Ng, weight = low bits), velocities single
The mesh side (deposit sums, Poisson solve, field) is float64 in every variant, as in the library
("fp32 push / fp64 Poisson"); the CIC deposit of the single-precision variants is the packed fixed-point one
(w_r rounded to 2^-24, w_l = 1 - w_r).  Nothing here is product code or the parity oracle: it is the error model
the bounds of tests/test_gpu_configs.py are derived from, and the GPU tests measure the same quantities.
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def yoshida():
    cbrt2 = 2.0 ** (1.0 / 3.0)
    w0 = (-1) * cbrt2 / (2 - cbrt2)
    w1 = 1 / (2 - cbrt2)
    return (0.5 * w1, 0.5 * (w0 + w1), 0.5 * (w0 + w1), 0.5 * w1), (0.0, w1, w0, w1)


def field_from_density(nsum, N, Ng, L, n0, ext):
    """Two-scan periodic solve of csrc/pic_solve.h on the raw weight sums."""
    dx = L / Ng
    b = nsum * (n0 * L / N / dx) - n0
    G = np.cumsum(b) * dx
    G -= G.mean()
    E = -0.5 * (G + np.roll(G, 1))
    return E if ext is None else E + ext


class Emu:
    def __init__(self, x0, v0, Ng, L, dt, xt, vt, wt):
        self.Ng, self.L, self.dt, self.N = Ng, L, dt, x0.size
        self.xt, self.vt, self.wt = xt, vt, wt
        self.dx = L / Ng
        if xt == "u32":
            self.x = np.rint(x0.astype(np.float64) / L * 2.0 ** 32).astype(np.uint64).astype(np.uint32)
        else:
            self.x = x0.astype(xt)
        self.v = v0.astype(vt)
        self.packed = (wt == np.float32) or xt == "u32"

    # -- locate: cell, right weight (in the weight type) ------------------------------------------
    def locate(self, q):
        Ng, L = self.Ng, self.L
        if self.xt == "u32":
            t = q.astype(np.uint64) * np.uint64(Ng)
            j = (t >> np.uint64(32)).astype(np.int64)
            frac = (t & np.uint64(0xFFFFFFFF)).astype(np.uint32)
            wr = frac.astype(np.float32) * np.float32(2.0 ** -32)
            return j, np.float32(1) - wr, wr, frac
        T = self.wt
        qq = q.astype(T)
        Lt, dxt = T(L), T(self.dx)
        xw = np.mod(np.mod(qq, Lt), Lt)
        jf = np.floor(xw / dxt)
        j = jf.astype(np.int64)
        j[j >= Ng] = 0
        wl = ((jf + T(1)) * dxt - xw) / dxt
        wr = (xw - jf * dxt) / dxt
        return j, wl, wr, None

    def deposit(self, q):
        j, wl, wr, frac = self.locate(q)
        Ng = self.Ng
        if self.packed:
            if frac is not None:
                f24 = np.minimum((frac.astype(np.uint64) + np.uint64(128)) >> np.uint64(8), np.uint64(1 << 24)).astype(np.float64)
            else:
                f24 = np.floor(np.clip(wr.astype(np.float32), 0, 1) * np.float32(1 << 24) + np.float32(0.5)).astype(np.float64)
            S = np.bincount(j, weights=f24, minlength=Ng) * 2.0 ** -24
            cnt = np.bincount(j, minlength=Ng).astype(np.float64)
            return cnt - S + np.roll(S, 1)
        n = np.bincount(j, weights=wl.astype(np.float64), minlength=Ng + 1)
        n += np.bincount(j + 1, weights=wr.astype(np.float64), minlength=Ng + 1)
        n[0] += n[Ng]
        return n[:Ng]

    def gather(self, q, E):
        j, wl, wr, _ = self.locate(q)
        T = np.float32 if self.xt == "u32" else self.wt
        Es = np.concatenate([E, E[:1]]).astype(T)
        return wl * Es[j] + wr * Es[j + 1]

    def drift(self, q, p, c):
        if self.xt == "u32":
            d = (np.float32(c) * p.astype(np.float32)) * np.float32(self.dt)
            du = np.rint(d * np.float32(2.0 ** 32 / self.L)).astype(np.int64)
            return ((q.astype(np.int64) + du) & 0xFFFFFFFF).astype(np.uint32)
        T = self.xt
        return q + (T(c) * p.astype(T)) * T(self.dt)

    def step(self, ext):
        cs, ds = yoshida()
        V = self.vt
        q, p = self.x, self.v
        for c, d in zip(cs, ds):
            if d != 0.0:
                E = field_from_density(self.deposit(q), self.N, self.Ng, self.L, 1.0, ext)
                Ep = self.gather(q, E).astype(V)
                p = p + (V(d) * (-Ep)) * V(self.dt)
            q = self.drift(q, p, c)
        if self.xt != "u32":
            T = self.xt
            q = np.mod(np.mod(q, T(self.L)), T(self.L))
        self.x, self.v = q, p

    def positions(self):
        if self.xt == "u32":
            return self.x.astype(np.float64) * (self.L / 2.0 ** 32)
        return self.x.astype(np.float64)

    def fields(self):
        nsum = self.deposit(self.x)
        E = field_from_density(nsum, self.N, self.Ng, self.L, 1.0, None)
        n = nsum * (1.0 * self.L / self.N / self.dx)
        ke = 0.5 * np.sum(self.v.astype(np.float64) ** 2)
        pe = 0.5 * np.sum(E * E) * self.dx * self.N / self.L
        return n, E, ke, pe


VARIANTS = {
    "f64": (np.float64, np.float64, np.float64),
    "f32": (np.float32, np.float32, np.float32),
    "x64": (np.float64, np.float32, np.float64),
    "v64": (np.float32, np.float64, np.float32),
    "w64": (np.float32, np.float32, np.float64),
    "u32": ("u32", np.float32, np.float32),
}


def run_variant(args):
    name, N, Ng, L, steps, seed, marks = args
    from oracle import pic_oracle as po
    from ocplasma_amd.control.actuator import E_field
    x0, v0 = po.synthetic_two_stream(N, L, seed=seed)
    x0 = x0.astype(np.float32).astype(np.float64)
    x0[x0 >= L] = 0.0
    v0 = v0.astype(np.float32).astype(np.float64)
    dt = min(0.1, 2 / np.sqrt(N / L))
    act = E_field(L, Ng, 3)
    rng = np.random.default_rng(seed)
    emu = Emu(x0, v0, Ng, L, dt, *VARIANTS[name])
    out = {}
    n, E, ke, pe = emu.fields()
    H0 = ke + pe
    for k in range(1, steps + 1):
        a = rng.uniform(-1.25, 1.25, (1, 6))
        emu.step(act.compute_E_batched(a)[0])
        if k in marks:
            n, E, ke, pe = emu.fields()
            out[k] = dict(x=emu.positions(), v=emu.v.astype(np.float64), n=n, E=E, ke=ke, pe=pe, H=ke + pe, H0=H0)
    return name, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--particles", type=int, default=1_000_000)
    ap.add_argument("--mesh", type=int, default=512)
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--seed", type=int, default=3)
    ap.add_argument("--procs", type=int, default=6)
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    L = 50.0
    marks = [k for k in (1, 10, 100, 500) if k <= args.steps]
    jobs = [(name, args.particles, args.mesh, L, args.steps, args.seed, marks) for name in VARIANTS]
    t0 = time.time()
    import multiprocessing as mp
    with mp.get_context("spawn").Pool(min(args.procs, len(jobs))) as pool:
        res = dict(pool.map(run_variant, jobs))
    ref = res["f64"]
    lines = [f"# float32 error model (CPU emulation of the sweep arithmetic; two-stream, N={args.particles}, "
             f"Ng={args.mesh}, dt={min(0.1, 2 / np.sqrt(args.particles / L)):.5f}, a new random action every step)", "",
             "Errors of each variant against the all-float64 emulation from the same start "
             "(max-norm, relative to the max of the reference quantity; x on the circle, relative to L).", "",
             "| variant (x / v / weights) | steps | x | v | n | E_mesh | KE | PE | H(t)/H(0)-1 variant | same, f64 |",
             "|---|---|---|---|---|---|---|---|---|---|"]
    label = {"f32": "f32 / f32 / f32", "x64": "f64 / f32 / f64", "v64": "f32 / f64 / f32", "w64": "f32 / f32 / f64",
             "u32": "u32 fixed / f32 / f32"}
    for name in ("f32", "x64", "v64", "w64", "u32"):
        for k in marks:
            a, b = res[name][k], ref[k]
            d = np.abs(a["x"] - b["x"])
            ex = np.max(np.minimum(d, L - d)) / L
            ev = np.max(np.abs(a["v"] - b["v"])) / np.max(np.abs(b["v"]))
            en = np.max(np.abs(a["n"] - b["n"])) / np.max(np.abs(b["n"]))
            eE = np.max(np.abs(a["E"] - b["E"])) / np.max(np.abs(b["E"]))
            lines.append(f"| {label[name]} | {k} | {ex:.1e} | {ev:.1e} | {en:.1e} | {eE:.1e} | {abs(a['ke'] / b['ke'] - 1):.1e} | "
                         f"{abs(a['pe'] / b['pe'] - 1):.1e} | {a['H'] / a['H0'] - 1:+.1e} | {b['H'] / b['H0'] - 1:+.1e} |")
    lines += ["", f"({time.time() - t0:.0f} s on {min(args.procs, len(jobs))} processes)"]
    text = "\n".join(lines)
    print(text)
    if args.out:
        with open(args.out, "w") as f:
            f.write(text + "\n")


if __name__ == "__main__":
    main()
