"""Coefficients of the Planck factor's 2^f in k_spectrum.hip (fast variant): (2^f - 1) / f on [-1/2, 1/2] by a polynomial of
degree 5, found by the Remez exchange on the RELATIVE error (mpmath, 40 digits).  Prints the coefficients as C literals and
the error reached; tests/test_spectrum_exp2.py holds the literals in the kernel source to this script's output.

    python tests/tools/exp2_coefficients.py
"""
import mpmath as mp

mp.mp.dps = 40
DEG = 5                      # degree of the polynomial for g(f) = (2^f - 1) / f; 2^f = 1 + f g(f) has degree 6
LN2 = mp.log(2)


def g(f):
    f = mp.mpf(f)
    if abs(f) < mp.mpf(10) ** -15:
        return LN2 + LN2 * LN2 * f / 2
    return mp.expm1(f * LN2) / f


def remez(deg, lo=mp.mpf(-0.5), hi=mp.mpf(0.5), rounds=12):
    n = deg + 2
    xs = [(lo + hi) / 2 + (hi - lo) / 2 * mp.cos(mp.pi * (n - 1 - k) / (n - 1)) for k in range(n)]
    for _ in range(rounds):
        # p(x_k) + (-1)^k E g(x_k) = g(x_k)     (relative error levelled)
        A = mp.matrix(n, n)
        b = mp.matrix(n, 1)
        for k, x in enumerate(xs):
            for j in range(deg + 1):
                A[k, j] = x ** j
            A[k, deg + 1] = (-1) ** k * g(x)
            b[k] = g(x)
        sol = mp.lu_solve(A, b)
        co = [sol[j] for j in range(deg + 1)]
        err = lambda x: mp.polyval(co[::-1], x) / g(x) - 1
        # new extrema: the largest |err| between consecutive sign changes on a fine grid
        grid = [lo + (hi - lo) * mp.mpf(i) / 4000 for i in range(4001)]
        vals = [err(x) for x in grid]
        ext, cur = [], 0
        for i in range(1, len(grid)):
            if mp.sign(vals[i]) != mp.sign(vals[cur]) and vals[i] != 0:
                seg = max(range(cur, i), key=lambda t: abs(vals[t]))
                ext.append(grid[seg])
                cur = i
        seg = max(range(cur, len(grid)), key=lambda t: abs(vals[t]))
        ext.append(grid[seg])
        if len(ext) != n:
            break
        xs = ext
    worst = max(abs(v) for v in vals)
    return co, worst


if __name__ == "__main__":
    co, worst = remez(DEG)
    print("relative error of 2^f - 1 on [-1/2, 1/2]: %.3e" % float(worst))
    for j, c in enumerate(co):
        print("C%d = %s" % (j + 1, mp.nstr(c, 20)))
    print("as doubles:", ", ".join(repr(float(c)) for c in co))
