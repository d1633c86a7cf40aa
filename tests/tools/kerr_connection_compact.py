"""The compact form of the Kerr connection (s5_kerr.hpp kerr_connection_compact, fast variant of the march kernel) against
(1) the Christoffel symbols of the Kerr metric, symbolically (sympy), and (2) the reference's own expressions (ref
src/sim5kerr.c:255-312, evaluated in long double) on 20 000 random points.  CPU only:  python tests/tools/kerr_connection_compact.py"""
import numpy as np
import sympy as sp

IDX = {'t01': (0, 0, 1), 't02': (0, 0, 2), 't13': (0, 1, 3), 't23': (0, 2, 3), 'r00': (1, 0, 0), 'r03': (1, 0, 3), 'r11': (1, 1, 1), 'r12': (1, 1, 2),
       'r22': (1, 2, 2), 'r33': (1, 3, 3), 'h00': (2, 0, 0), 'h03': (2, 0, 3), 'h11': (2, 1, 1), 'h12': (2, 1, 2), 'h22': (2, 2, 2), 'h33': (2, 3, 3),
       'p01': (3, 0, 1), 'p02': (3, 0, 2), 'p13': (3, 1, 3), 'p23': (3, 2, 3)}


def compact(a, r, c, s, sqrt=np.sqrt):
    """the twenty entries (off-diagonals doubled) as the device routine forms them; c = cos theta, s = sin theta"""
    c2 = c * c; s2 = s * s; cs = s * c; a2 = a * a; r2 = r * r; A2 = a2 + r2; a2c2 = a2 * c2; a2s2 = a2 * s2
    S = r2 + a2c2; D = A2 - 2 * r; w = r2 - a2c2
    inv = 1 / (S * D); S_1 = D * inv; S_2 = S_1 * S_1; S_3 = S_2 * S_1; DS2 = inv * S_1; SS = S * S; two_r = 2 * r; m_s = c / s
    G = {}
    G['t01'] = 2 * (A2 * w) * DS2
    G['t02'] = -4 * a2 * ((r * cs) * S_2)
    G['t13'] = (2 * a * s2) * ((a2c2 * (a2 - r2) - r2 * (a2 + 3 * r2)) * DS2)
    G['t23'] = -G['t02'] * (a * s2)
    G['r00'] = (D * w) * S_3
    G['r03'] = -2 * (a * s2) * G['r00']
    G['r11'] = (r * a2s2 - w) * inv
    G['r12'] = -2 * (a2 * cs) * S_1
    G['r22'] = -(r * D) * S_1
    G['r33'] = -(D * s2) * ((r * SS - a2s2 * w) * S_3)
    G['h00'] = -2 * (a2 * (r * cs)) * S_3
    G['h03'] = 4 * (a * (r * cs)) * (A2 * S_3)
    G['h11'] = (a2 * cs) * inv
    G['h12'] = two_r * S_1
    G['h22'] = -(a2 * cs) * S_1
    G['h33'] = -cs * ((A2 * SS + (a2s2 * two_r) * (A2 + S)) * S_3)
    G['p01'] = 2 * (a * w) * DS2
    G['p02'] = -4 * (a * r) * (m_s * S_2)
    G['p13'] = 2 * ((S * (r * S + a2s2) - 2 * (r2 * A2)) * DS2)
    G['p23'] = 2 * (m_s * ((SS + a2s2 * two_r) * S_2))
    return G


def reference_form(a, r, m):
    """ref src/sim5kerr.c:255-312, term for term"""
    rS = 2 * r; s = np.sqrt(1 - m * m); cs = s * m; c2 = m * m; s2 = s * s; cc = c2 - s2; CC = 8 * c2 * c2 - 8 * c2 + 1
    a2 = a * a; a4 = a2 * a2; a2cc = a2 * cc; a2c2 = a2 * c2; a2cs = a2 * cs; a4CC = a4 * CC; r2 = r * r; r3 = r2 * r; r4 = r2 * r2
    a2r2 = a2 * r2; a2_r2 = a2 + r2; Rq = a2 + 2 * r2 + a2cc; R = Rq * Rq; D = r2 - 2 * r + a2; S = r2 + a2c2
    S_1 = 1 / S; S_3 = 1 / (S * S * S); D_1 = 1 / D; R_1 = 1 / R; m_s = m / s; DR_1 = D_1 * R_1; DS_1 = D_1 * S_1; dbl_r2 = 2 * r2
    G = {}
    G['t01'] = 2 * 4 * a2_r2 * (r2 - a2c2) * DR_1; G['t02'] = 2 * -4 * a2cs * rS * R_1
    G['t13'] = 2 * 2 * a * s2 * (a4 - 3 * a2r2 - 6 * r4 + a2cc * (a2 - r2)) * DR_1; G['t23'] = -G['t02'] * s2 * a
    G['r00'] = D * (r2 - a2c2) * S_3; G['r03'] = -2 * G['r00'] * a * s2; G['r11'] = (r * (a2 - r) + a2 * (1 - r) * c2) * DS_1
    G['r12'] = -2 * a2cs * S_1; G['r22'] = -r * D * S_1
    G['r33'] = -D * s2 * (2 * a2c2 * r3 + r2 * r3 + a2 * a2c2 * s2 + a2c2 * a2c2 * r - a2r2 * s2) * S_3
    G['h00'] = -2 * r * a2cs * S_3; G['h03'] = 2 * -G['h00'] * a2_r2 / a; G['h11'] = a2cs * DS_1; G['h12'] = 2 * r * S_1; G['h22'] = -a2cs * S_1
    G['h33'] = -cs * (a2_r2 * S * S + a2 * s2 * rS * (a2_r2 + S)) * S_3
    G['p01'] = 2 * a * (r2 - a2c2) * DS_1 * S_1; G['p02'] = 2 * -4 * a * rS * m_s * R_1
    G['p13'] = (a4 + 3 * a4 * r - 12 * a2r2 + 8 * a2 * r3 - 16 * r4 + 8 * r2 * r3 + 4 * r * (dbl_r2 - r + a2) * a2cc - a4CC * (1 - r)) * DR_1
    G['p23'] = ((3 * a4 + 8 * a2 * r + 8 * a2r2 + 8 * r4 + 4 * (dbl_r2 - 2 * r + a2) * a2cc + a4CC) * m_s) * R_1
    return G


if __name__ == "__main__":
    r, a, th = sp.symbols('r a theta', positive=True)
    c, s = sp.cos(th), sp.sin(th)
    S = r**2 + a**2 * c**2; D = r**2 - 2 * r + a**2
    g = sp.Matrix([[-(1 - 2 * r / S), 0, 0, -2 * a * r * s**2 / S], [0, S / D, 0, 0], [0, 0, S, 0],
                   [-2 * a * r * s**2 / S, 0, 0, (r**2 + a**2 + 2 * a**2 * r * s**2 / S) * s**2]])
    X = [sp.Symbol('t'), r, th, sp.Symbol('phi')]
    gi = g.inv()
    Gc = compact(a, r, c, s)
    bad = 0
    for n, (i, j, k) in IDX.items():
        chris = sum(gi[i, l] * (sp.diff(g[l, j], X[k]) + sp.diff(g[l, k], X[j]) - sp.diff(g[j, k], X[l])) for l in range(4)) / 2
        d = sp.simplify(chris * (2 if j != k else 1) - Gc[n])
        print("%s: %s" % (n, "equals the Christoffel symbol" if d == 0 else "DIFFERS by %s" % d))
        bad += d != 0
    rng = np.random.default_rng(1)
    worst = {}
    for _ in range(20000):
        av = rng.uniform(1e-4, 0.9999); rv = (1 + np.sqrt(1 - av * av)) * 10 ** rng.uniform(0.02, 2.2); mv = rng.uniform(-0.999999, 0.999999)
        o = reference_form(np.longdouble(av), np.longdouble(rv), np.longdouble(mv))
        k = compact(av, rv, mv, np.sqrt(1 - mv * mv))
        for n in o:
            worst[n] = max(worst.get(n, 0.0), abs(float(k[n]) - float(o[n])) / max(abs(float(o[n])), 1e-300))
    print("worst relative difference from the reference's expressions (long double) over 20 000 points:", {n: "%.1e" % v for n, v in worst.items()})
    raise SystemExit(1 if bad or max(worst.values()) > 1e-10 else 0)
