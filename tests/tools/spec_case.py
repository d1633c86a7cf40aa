"""one job of tests/tools/fuzz_spectrum.py localised to a pixel: the local frame of the spectrum job by the batch (strict) routines
of the C-ABI -- the reference's tetrad chain, ref python/sim5diskraytrace.py:340-390 -- against the closed form of
k_spectrum.hip spectrum_stage_equatorial restated in numpy, for the pixels of one row   (python tests/tools/spec_case.py [row])"""
import sys, math, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import sim5_amd.capi as capi
from gpuutil import deg2rad
a, inc, nx, ny, rmax, hard, limb, dspin = 0.3, 10.180785690408381, 248, 88, 105.30801177850179, 1.6746061090489397, 0, 0.5488630162235405
y = int(sys.argv[1]) if len(sys.argv) > 1 else 57
kw = dict(y0=y, y1=y + 1, rmax=rmax, disk_spin=dspin, max_order=1, rms=1e-9)
S = capi.disk_image(capi.image_desc(nx, ny, a, deg2rad(inc), strict=True, **kw), full=True)
r = S["r"][0]; F = S["flux"][0]; hit = (S["cls"][0] == 2) & (F != 0)
ix = np.where(hit)[0]
alpha = ((ix + .5) / nx - .5) * 2 * rmax; beta = ((y + .5) / ny - .5) * 2 * rmax * (ny / nx)
incl = deg2rad(inc); l = -alpha * math.sin(incl); q = beta * beta + math.cos(incl) ** 2 * (alpha * alpha - a * a)
capi.disk_nt_setup(10.0, dspin, 0.1, 0.1)
ell = capi.disk_nt_ell(r[ix])
mt = capi.kerr_metric(a, r[ix], 0.0)
Om = capi.Omega_from_ell(ell, mt)
# strict chain: g = k_t / (k.U), U = A (1, 0, 0, Omega)
g00, g03, g33 = mt["g00"], mt["g03"], mt["g33"]
A2 = -(g00 + 2 * Om * g03 + Om * Om * g33)
g_chain = np.sqrt(A2) / (1.0 - Om * l)
# closed form of the fast kernel (its operands)
rms_d = float(np.float32(capi.disk_nt_r_min()))
rl = np.maximum(rms_d, r[ix]); x = np.sqrt(rl); af = float(np.float32(dspin))
Nl = rl * rl - 2 * af * x + af * af; Dl = x * rl - 2 * x + af
G00 = 2 - r[ix]; G03 = -2 * a; G33 = r[ix] * (r[ix] ** 2 + a * a) + 2 * a * a
No = -(G03 * Dl + Nl * G00); Do = G33 * Dl + Nl * G03
P = -(G00 * Do * Do + 2 * No * Do * G03 + No * No * G33)
den = np.where(Do < 0, -(Do - No * l), Do - No * l)
g_closed = np.sqrt(P / r[ix]) / den
T = (F[ix] / 5.670400e-05) ** 0.25
print("row", y, "hits", ix.size, "disk rms", rms_d, "r range", r[ix].min(), r[ix].max())
bad = np.where(~(np.abs(g_closed / g_chain - 1) < 1e-9))[0]
print("pixels where the closed form and the chain differ:", bad.size)
for j in bad[:10]:
    print("  px", ix[j], "r", r[ix][j], "l", l[j], "Omega", Om[j], "ell", ell[j], "A^-2", A2[j], "g chain", g_chain[j], "g closed", g_closed[j], "Do", Do[j], "den", (Do - No * l)[j], "T", T[j])
print("g<=0 or NaN in chain:", int((~(g_chain > 0)).sum()), " in closed form:", int((~(g_closed > 0)).sum()))
# which pixel: the difference fast - strict of the row's spectrum at energies across the peak, against every pixel's own contribution
E = np.array([1e-3, 0.03, 0.1, 0.3, 1.0, 3.0])
kw2 = dict(y0=y, y1=y + 1, rmax=rmax, disk_spin=dspin)
f = capi.disk_spectrum(capi.image_desc(nx, ny, a, deg2rad(inc), **kw2), E, hardening=hard, limb_darkening=limb)
s = capi.disk_spectrum(capi.image_desc(nx, ny, a, deg2rad(inc), strict=True, **kw2), E, hardening=hard, limb_darkening=limb)
d = f - s
h, kev2freq, c2, kB = 6.626069e-27, 2.417990e+17, 8.987554e+20, 1.380650e-16
contrib = np.zeros((ix.size, E.size))
for j in range(ix.size):
    Eg = E / g_chain[j]
    nu = kev2freq * Eg
    contrib[j] = 2.0 * h * nu ** 3 / c2 / hard ** 4 / np.expm1(h * kev2freq * Eg / (kB * hard * T[j])) * kev2freq * g_chain[j] ** 3
print("host sum / strict - 1:", contrib.sum(0) / s - 1)
print("fast - strict:", d, " relative", d / s)
shape = d / np.abs(d).max()
best = np.argsort([np.abs(c / np.abs(c).max() - np.abs(shape)).max() for c in contrib])[:3]
for j in best:
    print("  candidate px", ix[j], "r", r[ix][j], "T", T[j], "g", g_chain[j], "contribution / difference", contrib[j] / d)
