"""CPU: the emulated x87 arithmetic of golemflavor_amd/csrc/gf_x87.hpp (what the device uses to arbitrate the reference's
unitarity assert, fr.py:461-499) compiled for the host (tests/x87/x87_host.cpp, g++) and compared with the CPU's own x87
unit -- `long double` on x86-64, the format behind the reference's np.float128 / np.complex256 (fr.py:22-30) -- and, for
the whole chain, with the oracle, whose residual is the reference's bit for bit (test_oracle_golden.py G17).

The host build is test infrastructure: the product never evaluates this code on the CPU."""
import ctypes as C
import math
import os
import platform
import subprocess

import numpy as np
import pytest

from common import BIN_EDGES

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.skipif(platform.machine() != "x86_64", reason="needs the x87 unit as the reference")
LD = np.longdouble
Z = 1e-9
TEX = {1: (0.5, 1.0, Z, Z), 2: (Z, 0.25, Z, Z), 3: (Z, 1.0, 0.5, Z)}


@pytest.fixture(scope="module")
def x87lib(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("x87") / "libx87host.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-mfma", "-fPIC", "-shared", "-ffp-contract=off", "-o", out,
                           os.path.join(ROOT, "tests", "x87", "x87_host.cpp")])
    L = C.CDLL(out)
    L.x87t_bin_residual.restype = C.c_double
    L.x87t_pow10.argtypes = [C.c_uint64, C.c_int, C.c_double, C.c_double]
    L.x87t_bin_residual.argtypes = [C.POINTER(C.c_double)] * 2 + [C.c_double] * 4 + [C.c_int, C.c_void_p, C.c_void_p]
    return L


def _arr(x):
    return (C.c_double * len(x))(*[float(v) for v in x])


def test_basic_operations_are_bit_exact(x87lib):
    """+ - * / sqrt on 1.6 M operand pairs (random 64-bit significands, cancellations, small-integer factors -- the exact
    ties --, far-apart exponents, fp64 operands): every result equals the x87 unit's, bit for bit."""
    cnt = (C.c_int64 * 5)()
    for seed in (1, 2):
        x87lib.x87t_arith(C.c_uint64(seed), 800000, cnt)
        assert list(cnt) == [0, 0, 0, 0, 0], list(cnt)


def test_round4_primitives_equal_round3s(x87lib, tmp_path):
    """Round 4 shortened the two primitives every emulated + - * is made of (round64 without selects, the sum of two x87 numbers
    in 11 floating-point operations instead of 20; gf_x87.hpp).  Both forms on the same 48 M operand pairs -- random, cancelling,
    ties and near-ties 42-64 binades apart, sums under a power of two, short significands: the same digest of all results, and
    not one result that differs from the x87 unit's."""
    out = str(tmp_path / "libx87host_r3.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-mfma", "-fPIC", "-shared", "-ffp-contract=off", "-DGFX87_ROUND64_R3", "-o", out,
                           os.path.join(ROOT, "tests", "x87", "x87_host.cpp")])
    old = C.CDLL(out)
    for L in (x87lib, old):
        L.x87t_digest.restype = C.c_uint64
        L.x87t_digest.argtypes = [C.c_uint64, C.c_int64, C.POINTER(C.c_int64)]
    for seed in (5, 6, 7):
        w_new, w_old = C.c_int64(-1), C.c_int64(-1)
        d_new = x87lib.x87t_digest(seed, 16_000_000, C.byref(w_new))
        d_old = old.x87t_digest(seed, 16_000_000, C.byref(w_old))
        assert w_new.value == 0 and w_old.value == 0, (seed, w_new.value, w_old.value)
        assert d_new == d_old, seed


def test_round4_chain_residuals_equal_round3s(golden, x87lib, tmp_path):
    """The same A/B through the whole chain (angles_to_u with the emulated functions, both sandwiches, the Cardano chain, |X X^+|):
    the residual of 160 G17 rows x 20 energy bins, bit for bit the same double from both forms of the primitives."""
    out = str(tmp_path / "libx87host_r3.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-mfma", "-fPIC", "-shared", "-ffp-contract=off", "-DGFX87_ROUND64_R3", "-o", out,
                           os.path.join(ROOT, "tests", "x87", "x87_host.cpp")])
    old = C.CDLL(out)
    old.x87t_bin_residual.restype = C.c_double
    old.x87t_bin_residual.argtypes = x87lib.x87t_bin_residual.argtypes
    rows = golden["g17_rows"]
    centres = np.sqrt(BIN_EDGES[:-1] * BIN_EDGES[1:])
    rng = np.random.default_rng(3)
    n = big = 0
    for i in rng.permutation(len(rows))[:160]:
        r = rows[i]
        dim, tex, th = int(r[0]), int(r[1]), r[2:]
        sc2 = math.pow(10., th[6])
        for e in centres:
            a = x87lib.x87t_bin_residual(_arr(th[:4]), _arr(TEX[tex]), th[4], th[5], sc2, e, dim, None, None)
            b = old.x87t_bin_residual(_arr(th[:4]), _arr(TEX[tex]), th[4], th[5], sc2, e, dim, None, None)
            assert a == b or (a != a and b != b), (i, e, a, b)
            n += 1
            big += int(a > 1e-12)
    assert n == 3200 and big > 1000


def test_exact_shortcuts_of_the_three_lane_chain(golden, x87lib, tmp_path):
    """Round 4: the three-lane chain skips emulated arithmetic where the reference's own operations have an exact outcome -- z * z
    (the two cross products are one number, their sum its double), 2 z and -2 z, and z conj(z) (imaginary part exactly zero, so
    the diagonal of |X X^+| is a sum of squares and its own modulus).  The serial chain with the same shortcuts switched on
    (-DGFX87_SHORT_A/B/C) gives the same residual, bit for bit, on 300 G17 rows x 20 bins; on the device the three-lane chain is
    compared with the serial one (tests/test_gpu_unitarity_r3.py)."""
    out = str(tmp_path / "libx87host_short.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-mfma", "-fPIC", "-shared", "-ffp-contract=off", "-DGFX87_SHORT_A", "-DGFX87_SHORT_B",
                           "-DGFX87_SHORT_C", "-o", out, os.path.join(ROOT, "tests", "x87", "x87_host.cpp")])
    short = C.CDLL(out)
    short.x87t_bin_residual.restype = C.c_double
    short.x87t_bin_residual.argtypes = x87lib.x87t_bin_residual.argtypes
    rows = golden["g17_rows"]
    centres = np.sqrt(BIN_EDGES[:-1] * BIN_EDGES[1:])
    rng = np.random.default_rng(5)
    n = 0
    for i in rng.permutation(len(rows))[:300]:
        r = rows[i]
        dim, tex, th = int(r[0]), int(r[1]), r[2:]
        sc2 = math.pow(10., th[6])
        for e in centres:
            a = x87lib.x87t_bin_residual(_arr(th[:4]), _arr(TEX[tex]), th[4], th[5], sc2, e, dim, None, None)
            b = short.x87t_bin_residual(_arr(th[:4]), _arr(TEX[tex]), th[4], th[5], sc2, e, dim, None, None)
            assert a == b or (a != a and b != b), (i, e, a, b)
            n += 1
    assert n == 6000


def test_functions_are_correctly_rounded_neighbours_of_libm(x87lib):
    """asinl, acosl, sinl, cosl, hypotl: the emulation evaluates to ~2^-100 and rounds; glibc / the x87 microcode are
    faithful (< 1 ulp).  They must never differ by more than one unit in the last place, and agree in most calls."""
    ex, o1, w = (C.c_int64 * 5)(), (C.c_int64 * 5)(), (C.c_int64 * 5)()
    n = 100000
    x87lib.x87t_funcs(C.c_uint64(7), n, ex, o1, w)
    assert list(w) == [0] * 5, list(w)
    frac = np.array(list(ex)) / n
    assert np.all(frac > [0.85, 0.80, 0.95, 0.95, 0.99]), frac


def test_functions_are_correctly_rounded_against_mpmath(x87lib):
    """sin, cos, asin, acos to a 64-bit significand and 10^x to fp64: the emulation's results equal the CORRECTLY ROUNDED values
    (mpmath at 300 bits, rounded to nearest even) on every one of 3 000 random 64-bit arguments -- the definition the chain
    relies on (a ~2^-100 evaluation rounds wrongly only within ~2^-36 ulp of a tie), and the pin of round 3's faster series
    (Horner's rule with reciprocal-factorial constants instead of term-by-term divisions)."""
    import mpmath as mp
    mp.mp.prec = 300
    x87lib.x87t_eval.argtypes = [C.c_double, C.c_double, C.POINTER(C.c_double)]
    x87lib.x87t_pow10_value.restype = C.c_double
    x87lib.x87t_pow10_value.argtypes = [C.c_double]

    def round_to(v, bits):
        if v == 0:
            return mp.mpf(0)
        e = int(mp.floor(mp.log(abs(v), 2)))
        q = mp.mpf(2) ** (e - bits + 1)
        return mp.nint(v / q) * q                     # mp.nint rounds half to even

    rng = np.random.default_rng(11)
    out = (C.c_double * 8)()
    bad = {"sin": 0, "cos": 0, "asin": 0, "acos": 0, "pow10": 0}
    for it in range(3000):
        # a random 64-bit significand as (hi, lo)
        m = int(rng.integers(1 << 63, (1 << 64) - 1, dtype=np.uint64))
        scale = [2.0 ** -61, 2.0 ** -64, 2.0 ** -74][it % 3]                     # arguments in [4, 8), [0.5, 1), [2^-11, 2^-10)
        v = mp.mpf(m) * mp.mpf(scale) * (-1 if it % 5 == 0 else 1)
        hi = float(v)
        lo = float(v - mp.mpf(hi))
        assert mp.mpf(hi) + mp.mpf(lo) == v
        x87lib.x87t_eval(hi, lo, out)
        got = [mp.mpf(out[2 * k]) + mp.mpf(out[2 * k + 1]) for k in range(4)]
        bad["sin"] += int(got[0] != round_to(mp.sin(v), 64))
        bad["cos"] += int(got[1] != round_to(mp.cos(v), 64))
        if abs(v) <= 1:
            bad["asin"] += int(got[2] != round_to(mp.asin(v), 64))
            bad["acos"] += int(got[3] != round_to(mp.acos(v), 64))
        x = float(rng.uniform(-72.0, -20.0))
        bad["pow10"] += int(mp.mpf(x87lib.x87t_pow10_value(x)) != round_to(mp.mpf(10) ** mp.mpf(x), 53))
    assert bad == {"sin": 0, "cos": 0, "asin": 0, "acos": 0, "pow10": 0}, bad


def test_reciprocal_constants_are_the_divisions_they_replace(x87lib):
    """1/2, 1/3, 1/9, 1/54 as 64-bit-significand constants (fr.py:205-209's divisions of constants, four per energy bin) equal
    the emulated divisions AND the CPU's own long-double quotients."""
    assert x87lib.x87t_consts() == 0


def test_pow10_is_libms(x87lib):
    """fr.py:380 np.power(10., logLam): the chain's one fp64 transcendental.  The emulation's correctly rounded 10^x
    equals libm's pow (correctly rounded on all but ~0.1 % of arguments) on >= 99.7 % of the scale range."""
    n = 200000
    bad = x87lib.x87t_pow10(C.c_uint64(3), n, C.c_double(-72.0), C.c_double(-20.0))
    assert bad / n < 0.003, bad / n


def test_chain_residual_vs_oracle(golden, oracle, x87lib):
    """The whole chain (fr.py:380-399 + 489-494) per (walker, bin) on the G17 sweep, against the oracle's per-bin
    residual (= the reference's).  With the mixing matrices handed over in long double (what gf_model_create does for
    per-model constants) only cacosl / ccosl / hypotl can differ: nearly every pair is bit-identical.  With the SM matrix
    built by the emulated asin / acos / sin / cos (sampled angles) about half of the matrices differ from libm's in a
    last bit; the residual of the pairs that matter (> 1e-12) stays within a factor two."""
    LO = oracle.lib()
    rows = golden["g17_rows"]
    centres = np.sqrt(BIN_EDGES[:-1] * BIN_EDGES[1:])
    rng = np.random.default_rng(0)
    tot = 0
    exact = {"host": 0, "emu": 0}
    worst = {"host": 0.0, "emu": 0.0}
    nbig = 0
    verdict_diff = 0
    for i in rng.permutation(len(rows))[:160]:
        r = rows[i]
        dim, tex, th = int(r[0]), int(r[1]), r[2:]
        sc2 = math.pow(10., th[6])
        smu, npu = np.zeros(18, dtype=LD), np.zeros(18, dtype=LD)
        LO.orc_angles_to_u_ldout(_arr(th[:4]), smu.ctypes.data_as(C.c_void_p))
        LO.orc_angles_to_u_ldout(_arr(TEX[tex]), npu.ctypes.data_as(C.c_void_p))
        for e in centres:
            out = np.zeros(96, dtype=LD)
            LO.orc_debug_bsmu_ld(_arr(TEX[tex]), C.c_double(th[6]), dim, C.c_double(e), _arr(th[4:6]), _arr(th[:4]),
                                 out.ctypes.data_as(C.c_void_p))
            ro = float(out[54])
            ra = x87lib.x87t_bin_residual(_arr(th[:4]), _arr(TEX[tex]), th[4], th[5], sc2, e, dim,
                                          smu.ctypes.data_as(C.c_void_p), npu.ctypes.data_as(C.c_void_p))
            rb = x87lib.x87t_bin_residual(_arr(th[:4]), _arr(TEX[tex]), th[4], th[5], sc2, e, dim, None,
                                          npu.ctypes.data_as(C.c_void_p))
            tot += 1
            exact["host"] += int(ra == ro)
            exact["emu"] += int(rb == ro)
            if ro > 1e-12:
                nbig += 1
                worst["host"] = max(worst["host"], abs(math.log10(ra / ro)))
                worst["emu"] = max(worst["emu"], abs(math.log10(rb / ro)))
            verdict_diff += int((rb >= 1e-7) != (ro >= 1e-7))
    assert tot == 3200 and nbig > 1000
    assert exact["host"] / tot > 0.95 and exact["emu"] / tot > 0.75, exact
    assert worst["host"] < 0.35 and worst["emu"] < 0.35, worst          # a factor ~2 at most
    assert verdict_diff <= 2
