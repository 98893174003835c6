#!/usr/bin/env python3
"""VERDICT r5 item 5, on paper first: how accurate can a PACKED-FLOAT16 polynomial tanh be (v_pk_fma_f16 Horner on a clamped, normalised argument - the
form that would replace the activation's two quarter-rate transcendentals)?  Least-squares / reweighted minimax fit of tanh(X s) = s P(s^2) on |s| <= 1 and
its evaluation in simulated float16 arithmetic (every operation rounded to float16, fused multiply-adds rounded once), against the product form
(float32 exp2 / rcp, ONE rounding to float16 at the end).  NumPy only."""
import numpy as np
f16 = np.float16


def fit(X, deg_t):
    s = np.cos(np.linspace(0, np.pi, 4001)); s = s[s > 1e-9]; t = s * s
    A = np.stack([s * t ** k for k in range(deg_t + 1)], 1); y = np.tanh(X * s); w = np.ones_like(s)
    for _ in range(60):
        c = np.linalg.lstsq(A * w[:, None], y * w, rcond=None)[0]
        e = np.abs(A @ c - y); w = w * (1 + 4 * e / e.max()); w /= w.mean()
    return c, np.abs(A @ c - y).max()


def eval16(c, X, a32, cpre):
    k = f16(1.0 / (cpre * X) / 2)
    a16 = a32.astype(f16)
    u = np.clip(a16.astype(np.float64) * np.float64(k) + 0.5, 0, 1).astype(f16)          # v_pk_fma_f16 ... clamp
    s = (u.astype(np.float64) * 2 - 1).astype(f16)
    t = (s.astype(np.float64) ** 2).astype(f16)
    c16 = [f16(x) for x in c]
    p = np.full(a32.shape, c16[-1], dtype=f16)
    for ck in c16[-2::-1]:
        p = (p.astype(np.float64) * t.astype(np.float64) + np.float64(ck)).astype(f16)
    return (p.astype(np.float64) * s.astype(np.float64)).astype(f16)


cpre = 2.8853900817779268
z = np.linspace(-8, 8, 400001); ref = np.tanh(z)
cur = (1 - 2 / (np.exp2((cpre * z).astype(np.float32)) + 1)).astype(np.float32).astype(f16)
print(f"product form (float32 exp2 / rcp, one rounding to float16): max |error| {np.abs(cur.astype(np.float64) - ref).max():.2e}")
print("packed-float16 odd polynomial on a clamped argument (instructions per PAIR of activations: convert, affine + clamp, re-centre, square, Horner, final multiply):")
for X in (2.5, 3.0, 3.5, 4.0):
    for d in (2, 3, 4):
        c, e = fit(X, d)
        err = np.abs(eval16(c, X, (cpre * z).astype(np.float32), cpre).astype(np.float64) - ref)
        print(f"  clamp at |z| = {X}, odd degree {2 * d + 1} ({5 + d} packed instructions per pair): fit error {e:.1e}, evaluated in float16 max |error| {err.max():.1e}, "
              f"for |z| < 0.1 {err[np.abs(z) < 0.1].max():.1e}")
