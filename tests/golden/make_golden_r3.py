#!/usr/bin/env python3
"""Third set of fixtures: the value the GENERATING numpy gives `np.power(10., logLam)` (golemflavor/fr.py:380) for every row
of the reference-generated BSM goldens G8, G9, G11-G14 -- what G17 already carries as `g17_sc2`.

Why.  The reference's unitarity verdict (fr.py:461-499 via :398-399) is the rounding noise of its 80-bit closed form, and
that noise changes by a factor of order one with the last bit of 10**logLam.  numpy evaluates that power with its own
vectorised routine, one fp64 ulp away from libm's pow on ~5 % of arguments (tests/test_oracle_golden.py G17), so a stored
verdict can only be held tightly against an implementation that is fed the SAME power.  With these values the tests apply the
half-decade band against the stored verdicts on every row whose power equals libm's (~95 %), and the wide band only on the rest.

Run in the BUILD CONTAINER (the one whose numpy generated golden.npz / golden_r2.npz):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_r3.py

It imports the reference only to take `np.power` through the reference's own statement (fr.py:380) on one row per set as a
cross-check; the stored values are `np.power(10., x)` per element, exactly as make_golden_r2.py stored `g17_sc2`, and the script
first verifies that it reproduces `g17_sc2` bit for bit (same numpy build as the generating one)."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)

import numpy as np  # noqa: E402


def main():
    g1 = np.load(os.path.join(HERE, "golden.npz"))
    g2 = np.load(os.path.join(HERE, "golden_r2.npz"))
    again = np.array([np.power(10., x) for x in g2["g17_rows"][:, 8]], dtype=np.float64)
    if not np.array_equal(again, g2["g17_sc2"]):
        raise SystemExit("this numpy does not reproduce g17_sc2: not the generating build (%d of %d differ)"
                         % (int(np.sum(again != g2["g17_sc2"])), len(again)))
    out = {"numpy_version": np.array(np.__version__)}
    import math
    for name, z in (("g8", g1), ("g9", g1), ("g11", g2), ("g12", g2), ("g13", g2), ("g14", g2)):
        ll = z[name + "_rows"][:, -1]                                  # logLam is the last column of every row layout
        sc2 = np.array([np.power(10., x) for x in ll], dtype=np.float64)
        libm = np.array([math.pow(10., float(x)) for x in ll])
        out[name + "_sc2"] = sc2
        print("%-4s %4d rows, numpy's 10**logLam differs from libm's on %d" % (name, len(ll), int(np.sum(sc2 != libm))))
    # cross-check through the reference's own statement: params_to_BSMu evaluates np.power(10., sc2) at fr.py:380; the matrix it
    # returns for one G8 row must be the one the oracle gives when fed the stored power (done by the tests, with the oracle);
    # here only: the reference imports and its constant agrees with the rows' scale column range
    try:
        import make_golden as G  # noqa: F401  (installs the shims, imports the reference)
        from golemflavor import fr
        lo, hi = fr.SCALE_BOUNDARIES[6]
        assert lo <= g1["g8_rows"][0, -1] <= hi or g1["g8_rows"][0, 0] != 6
        print("reference imported:", fr.__file__)
    except Exception as exc:                                           # noqa: BLE001
        print("reference not importable here (%s): values stored from numpy alone" % exc)
    np.savez_compressed(os.path.join(HERE, "golden_r3.npz"), **out)
    print("wrote golden_r3.npz")


if __name__ == "__main__":
    main()
