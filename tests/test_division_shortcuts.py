"""CPU: the arithmetic identities the kernels' division shortcuts rest on (q1physrl_amd/csrc/q1env_device.hpp div_const1, observe<float>),
checked without a GPU - exhaustively where the operand set is finite, in exact rational arithmetic otherwise.  The on-device twin is
q1env_selftest_division (tests/test_hip_parity.py::test_exact_division_shortcuts_selftest).

Reference expressions: env.py:381-400 (`_round_vel` = trunc(vel / 16) * 16, `_round_origin` = round(z * 8) / 8, obs / get_obs_scale in
float64; the float32 row is DEFINED as that row rounded to float32), env.py:236 and phys.py:59 (true divisions by action_range, 180)."""
import random
from fractions import Fraction as F

import numpy as np


def test_z_column_one_product_is_the_reference_value_for_every_j():
    j = np.arange(0, 1 << 24, dtype=np.float64)                     # rint(8 z): every value below 2^24
    ref = ((j * 0.125) / 100.0).astype(np.float32)                  # RN32(RN64(RN64(j / 8) / 100))
    new = (j * (1.0 / 800.0)).astype(np.float32)                    # RN32(RN64(j * RN64(1 / 800)))
    assert np.array_equal(ref.view(np.uint32), new.view(np.uint32))


def test_vel_columns_one_step_float32_division_is_the_reference_value_for_every_m():
    m = np.arange(-(1 << 24) + 1, 1 << 24, dtype=np.float64)        # trunc(v / 16): every value below 2^24 in magnitude
    ref = ((m * 16.0 + 0.0) / 200.0).astype(np.float32)             # RN32(RN64(16 m / 200))
    y = np.float64(np.float32(0.08))
    # float32 FMA chain emulated EXACTLY in float64: m y has <= 48 significant bits, the residual m - 12.5 q0 is a short cancellation,
    # and q0 + r y spans <= 53 bits, so each float64 operation below is exact and .astype(float32) is the single rounding of the fma
    q0 = (m * y).astype(np.float32).astype(np.float64)
    r = m - 12.5 * q0
    assert np.array_equal(r.astype(np.float32).astype(np.float64), r)
    q1 = (q0 + r * y).astype(np.float32)
    assert np.array_equal(ref.view(np.uint32), q1.view(np.uint32))
    assert abs(F(float(np.float32(0.08))) * F(25, 2) - 1) <= F(1, 2 ** 25)      # the float32 one-step bound


def _one_step(x, c):
    y = 1.0 / c                                                     # RN(1 / c)
    q0 = float(F(x) * F(y))                                         # float(Fraction) rounds to nearest even: RN(x y)
    r = F(x) - F(q0) * F(c)                                         # exact (what the fma computes before its rounding)
    return float(F(q0) + r * F(y))                                  # RN(q0 + r y)


def test_one_step_constant_division_is_correctly_rounded_when_the_reciprocal_bound_holds():
    rnd = random.Random(5)
    consts = [180.0, 90.0, 10.0, float(np.float32(720) * np.float32(0.014)), 100.0, 200.0]
    for c in consts:
        e = abs(F(1.0 / c) * F(c) - 1)
        assert e <= F(1, 2 ** 54), (c, float(e * 2 ** 54))          # |c RN(1/c) - 1| <= 2^-54: RN(x y) is faithful, Markstein applies
        xs = [rnd.uniform(-2000, 2000) for _ in range(3000)] + [rnd.uniform(0, 10) for _ in range(3000)]
        xs += [float(np.nextafter(c * k, s)) for k in range(1, 120) for s in (-np.inf, np.inf)] + [float(k) * c for k in range(1, 120)]
        xs += [rnd.uniform(1, 2) * 2.0 ** rnd.randint(-60, 60) for _ in range(3000)]
        for x in xs:
            assert _one_step(x, c) == float(F(x) / F(c)), (x, c)


def test_the_bound_is_what_separates_one_step_from_two():
    """A constant whose reciprocal misses the bound must be refused by the host-side test the SPEC kernels use (q1env_host.hpp
    div_one_step_ok); this restates that test and shows both outcomes occur."""
    def ok(c):
        return abs(F(1.0 / c) * F(c) - 1) <= F(1, 2 ** 54)
    assert ok(10.0) and ok(10.079999923706055) and ok(180.0)
    assert any(not ok(float(k) / 7.0) for k in range(1, 200))
