"""Randomised parity campaign for the image kernels (run on the GPU box; not part of the suite):
  fast(mirror) == fast(plain halves) bit for bit;  fast vs strict: same classes, r and g within 1e-7; flux of either variant
  within 1e-6 of the checker's disk_nt_flux at the variant's own radii, NO floor;  strict vs the CPU oracle: same classes, r within 1e-9.
Since round 5 the central column of an odd-width image (alpha = 0 exactly) and the central row of an odd-height one (beta = 0)
are in the class comparisons like every other pixel.
usage: python tests/tools/fuzz_images.py [n_cases] [seed]"""
import sys, math, time, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import sim5_amd.capi as capi
from gpuutil import deg2rad
import oraclelib as ol

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
t0 = time.time()
for case in range(ncases):
    a = float(rng.choice([0.0, 1e-5, 0.3, 0.7, 0.9, 0.998, 0.9999, rng.uniform(0, 0.999)]))
    inc = float(rng.uniform(3.0, 87.0))
    nx, ny = int(rng.integers(17, 300)), int(rng.integers(2, 300))
    order = int(rng.choice([1, 2]))
    rmax = float(rng.choice([0.0, rng.uniform(3.0, 60.0)]))
    mk = lambda lo, hi, strict=False: capi.disk_image(capi.image_desc(nx, ny, a, deg2rad(inc), y0=lo, y1=hi, max_order=order,
                                                                   rmax=rmax, strict=strict), full=True)
    sym = mk(0, ny)
    # the production instantiation (no full-precision planes: the job-list kernel): its two f32 planes, bit for bit
    prod = capi.disk_image(capi.image_desc(nx, ny, a, deg2rad(inc), max_order=order, rmax=rmax), full=False)
    cut = ny // 2 + 1 if ny > 2 else 1
    top, bot = mk(0, cut), mk(cut, ny) if cut < ny else None
    msg = []
    for k in ("image_f", "image_g"):
        if not np.array_equal(prod[k].view(np.uint32), sym[k].view(np.uint32)):
            msg.append("production kernel != aux kernel in %s (%d px)" % (k, int((prod[k].view(np.uint32) != sym[k].view(np.uint32)).sum())))
    for k in ("cls", "gtype", "image_f", "image_g", "r", "g", "flux"):
        both = top[k] if bot is None else np.concatenate([top[k], bot[k]], axis=0)
        if not np.array_equal(sym[k], both, equal_nan=True):
            msg.append("mirror!=plain in %s (%d px)" % (k, int((~((sym[k] == both) | (np.isnan(sym[k].astype(float)) & np.isnan(both.astype(float))))).sum())))
    st = mk(0, ny, strict=True)
    # alpha = 0 exactly on the central column of an odd-width image: l = 0, a degenerate quartic whose class the REFERENCE
    # itself decides by rounding noise (any libm, any operation order gives another pattern there) -- left out, counted
    # (round 5: the central column and row are compared like every other pixel -- the x87 roundings of the reference's polar
    # roots that decide their classes are reproduced; VALUES on the central row stay out: with beta = 1e-6 the
    # observer sits on the polar turning point and Tip cancels to rounding noise)
    col = np.ones((ny, nx), bool)
    val = col.copy()
    if ny % 2 == 1:
        val[ny // 2, :] = False
    note = ""
    note_in = ""
    if (st["cls"] != sym["cls"])[~col].any():
        note = " [central column / row: %d px differ]" % int((st["cls"] != sym["cls"])[~col].sum())
    if not np.array_equal(st["cls"][col], sym["cls"][col]):
        msg.append("fast/strict classes differ at %d px" % int((st["cls"] != sym["cls"])[col].sum()))
    same = (st["cls"] == sym["cls"]) & np.isfinite(st["r"]) & val
    if same.any():
        er = np.abs(sym["r"][same] / st["r"][same] - 1).max(); eg = np.abs(sym["g"][same] - st["g"][same]).max()
        # flux, no floor (round 6): each variant against the checker's disk_nt_flux at ITS OWN radii -- the same input bits
        # (tests/gpuutil.py assert_flux); ef = the worse of the two
        ef = 0.0
        for v in (sym, st):
            own = ol.cpu_disk_flux(v["r"][same], a)
            with np.errstate(divide="ignore", invalid="ignore"):
                e = np.where(v["flux"][same] == own, 0.0, np.abs(v["flux"][same] - own) / np.abs(own))
            ef = max(ef, float(e.max()))
        fl = np.maximum(np.abs(st["flux"][same]), 1e-300)
        explained = ""
        if er > 1e-9 or ef > 1e-6:
            # is it the INPUT?  the fast variant's pixel coordinates differ from the reference's expression by an ulp (rows iy and
            # ny-1-iy get exactly opposite beta); the checker itself, run on the worst pixel with alpha / beta one unit in the last
            # place away, says how far the reference's own result moves for that
            er_map = np.where(same, np.abs(sym["r"] / np.where(same, st["r"], 1.0) - 1), 0.0)
            wy, wx = np.unravel_index(int(np.argmax(er_map)), er_map.shape)
            z1 = 1 + (1 - a * a) ** (1 / 3) * ((1 + a) ** (1 / 3) + (1 - a) ** (1 / 3)); z2 = math.sqrt(3 * a * a + z1 * z1)
            rms = 3 + z2 - math.sqrt((3 - z1) * (3 + z1 + 2 * z2))
            orc = ol.Oracle(); orc.disk_nt_setup(10.0, a, 0.1, 0.1, 0)
            rm_ = rmax if rmax > 0.0 else rms + 8.0
            al0 = ((wx + .5) / nx - 0.5) * 2.0 * rm_; be0 = ((wy + .5) / ny - 0.5) * 2.0 * rm_ * (ny / nx)
            r0 = orc.disk_pixel(deg2rad(inc), a, rms, al0, be0).r
            # (TWO units for beta: the fast variant gives a row of the lower half minus the value of its mirror row, and the two
            # quotients can round apart by two -- case 1610 of the 40 000-job campaign of round 6, profiles/r06_fuzz_summary.txt)
            def ulps(x, k):
                for _ in range(abs(k)):
                    x = float(np.nextafter(x, x + k))
                return x
            moved = max(abs(orc.disk_pixel(deg2rad(inc), a, rms, ulps(al0, da), ulps(be0, db)).r / r0 - 1)
                        for da, db in ((0, 1), (0, -1), (1, 0), (-1, 0), (0, 2), (0, -2)))
            if moved >= 0.5 * er_map[wy, wx]:
                explained = " [pixel (%d,%d): the CHECKER's r moves by %.1e for one or two ulp of alpha / beta -- the difference is the input's]" % (wy, wx, moved)
            elif 2 * wy + 1 > ny and (ny & (ny - 1)) != 0:
                # a row of the lower half of a height that is not a power of two: the fast variant's beta is MINUS the value of the
                # mirror row (s5_thindisk.hpp pixel_beta).  (iy + .5) / ny - 0.5 cancels, so the two rows' values round apart by up to
                # 2^-53 / |iy / ny - 0.5| relative -- a few to a few hundred units in the last place.  The checker AT THAT beta:
                be_img = -((((ny - 1 - wy) + .5) / ny - 0.5) * 2.0 * rm_ * (ny / nx))
                r_own = orc.disk_pixel(deg2rad(inc), a, rms, al0, be_img).r
                if abs(r_own / sym["r"][wy, wx] - 1) < 1e-9:
                    explained = (" [pixel (%d,%d): at the fast variant's own beta, %d ulp from the reference's, the CHECKER has the fast variant's r to %.1e"
                                 " and its r at the reference's beta is %.1e away: a step of the reference's own r(beta)]" % (
                                     wy, wx, int(round((be_img - be0) / np.spacing(be0))), abs(r_own / sym["r"][wy, wx] - 1), abs(r_own / r0 - 1)))
        if explained:
            note_in = explained
        elif er > 1e-7 or eg > 1e-7 or ef > 1e-6:
            # where: radius of the worst flux pixel, its distance from the inner edge of the flux (x - x0 in x = sqrt(r)), the two values
            e_all = np.abs(sym["flux"][same] - st["flux"][same]) / fl
            j = int(np.argmax(e_all))
            rw = st["r"][same][j]
            lit = st["r"][same][st["flux"][same] > 0]
            msg.append("fast vs strict r %.1e g %.1e flux %.1e [worst flux pixel: r %.9f, sqrt(r) - sqrt(r_in) %.2e, F fast %.6e strict %.6e, F/peak %.1e]" % (
                er, eg, ef, rw, math.sqrt(rw) - math.sqrt(lit.min()) if lit.size else float("nan"), sym["flux"][same][j], st["flux"][same][j],
                st["flux"][same][j] / np.abs(st["flux"]).max()))
    if nx * ny <= 40000:
        c = ol.cpu_disk_image("port", nx, ny, a, inc, nthreads=8, full=True) if (order == 2 and rmax == 0.0) else None
        if c is not None:
            if not np.array_equal(c["cls"][col], st["cls"][col]):
                msg.append("strict/oracle classes differ at %d px" % int((c["cls"] != st["cls"])[col].sum()))
            ok = (c["cls"] == st["cls"]) & np.isfinite(c["r"]) & val
            if ok.any() and np.abs(st["r"][ok] / c["r"][ok] - 1).max() > 1e-9:
                msg.append("strict vs oracle r %.1e" % np.abs(st["r"][ok] / c["r"][ok] - 1).max())
    print("case %3d a=%.6g inc=%.2f %dx%d order=%d rmax=%.3g hits=%d : %s" % (case, a, inc, nx, ny, order, rmax, int(np.isfinite(sym["r"]).sum()),
                                                                          ("ok" if not msg else "; ".join(msg)) + note + note_in), flush=True)
    bad += bool(msg)
print("%d cases, %d with findings, %.0f s" % (ncases, bad, time.time() - t0))
sys.exit(1 if bad else 0)
