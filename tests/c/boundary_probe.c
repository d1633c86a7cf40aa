/* boundary_probe.c -- a C program against the SIM5 scalar API (sim5_amd/host/sim5lib.h) that calls the public
 * prototypes which are NOT on the inner ray-tracing path (flat-space metric and connection for an RTOPT_FLAT caller,
 * Gamma, vector helpers, general / radial tetrads, four-velocities, epicyclic frequencies, Legendre integrals by angle,
 * black-body spectrum) and prints one "name v0 v1 ..." line per call; tests/test_gpu_host_shim.py compares every
 * number with the unmodified reference called with the same arguments.
 *   usage: boundary_probe <spin> <r> <m>
 */
#include "sim5lib.h"

static void line(const char *name, const double *v, int n)
{
    printf("%s", name);
    for (int i = 0; i < n; i++) printf(" %.17g", v[i]);
    printf("\n");
}

int main(int argc, char **argv)
{
    if (argc != 4) { fprintf(stderr, "usage: %s spin r m\n", argv[0]); return 2; }
    const double a = atof(argv[1]), r = atof(argv[2]), m = atof(argv[3]);
    sim5metric g, gc, gf, gfc;
    kerr_metric(a, r, m, &g);
    kerr_metric_contravariant(a, r, m, &gc);
    flat_metric(r, m, &gf);
    flat_metric_contravariant(r, m, &gfc);
    line("kerr_metric_contravariant", &gc.g00, 5);
    line("flat_metric", &gf.g00, 5);
    line("flat_metric_contravariant", &gfc.g00, 5);

    /* an RTOPT_FLAT caller: a null vector in flat space, its acceleration from flat_connection + Gamma */
    double G[4][4][4], k[4], dk[4], kc[4];
    flat_connection(r, m, G);
    vector_set(k, 1.0, -0.6, 0.02, 0.01);
    vector_norm_to_null(k, 1.0, &gf);
    Gamma(G, k, k, dk);
    line("flat_null_k", k, 4);
    line("flat_Gamma", dk, 4);
    { double kk = dotprod(k, k, &gf); line("flat_kk", &kk, 1); }
    kerr_connection(a, r, m, G);
    vector_copy(k, kc);
    vector_norm_to_null(kc, 2.0, &g);
    Gamma(G, kc, kc, dk);
    line("kerr_null_k", kc, 4);
    line("kerr_Gamma", dk, 4);
    vector_covariant(kc, dk, &g);
    line("kerr_k_cov", dk, 4);
    vector_covariant(kc, dk, NULL);
    line("flat_k_cov", dk, 4);
    {
        double sp[4] = { 0.0, 0.3, -0.2, 0.05 }, v[3];
        v[0] = vector_norm(sp, &g); v[1] = vector_norm(sp, NULL); v[2] = vector_3norm(sp);
        vector_multiply(sp, 2.5);
        line("norms", v, 3);
        line("multiplied", sp, 4);
    }

    /* observers */
    double U[4];
    sim5tetrad t;
    const double Om = 0.7 * OmegaK(r, a);
    fourvelocity_zamo(&g, U);            line("fourvelocity_zamo", U, 4);
    fourvelocity_azimuthal(Om, &g, U);   line("fourvelocity_azimuthal", U, 4);
    fourvelocity_radial(-0.2, &g, U);    line("fourvelocity_radial", U, 4);
    { double N = fourvelocity_norm(0.05, 0.01, 0.5 * Om, &g); line("fourvelocity_norm", &N, 1); }
    fourvelocity(0.05, 0.01, 0.5 * Om, &g, U); line("fourvelocity", U, 4);
    tetrad_general(&g, U, &t);           line("tetrad_general", &t.e[0][0], 16);
    tetrad_radial(&g, -0.2, &t);         line("tetrad_radial", &t.e[0][0], 16);
    tetrad_radial(&g, 0.0, &t);          line("tetrad_radial0", &t.e[0][0], 16);
    {
        double v[3];
        v[0] = omega_r(r + 6.0, a); v[1] = omega_z(r + 6.0, a); v[2] = ell_from_Omega(Om, &g);
        line("frequencies", v, 3);
    }

    /* geodesic: sign of k^theta next to dm_sign */
    {
        geodesic gd; int err = 0;
        double v[3] = { NAN, NAN, NAN };
        if (geodesic_init_inf(deg2rad(65.0), a, 4.0, -3.0, &gd, &err)) {
            v[0] = geodesic_position_pol_sign_k_theta(&gd, 0.4 * gd.Rpc);
            v[1] = geodesic_dm_sign(&gd, 0.4 * gd.Rpc);
            v[2] = geodesic_position_pol_sign_k_theta(&gd, 1.7 * gd.Rpc);
        }
        line("sign_k_theta", v, 3);
    }

    /* Legendre integrals by angle / sine */
    {
        double v[7];
        sim5complex z = elliptic_pi(-2.2, 1.8, 0.45), w = elliptic_pi(4.0, -0.6, 0.45);
        v[0] = elliptic_f(-2.2, 0.45); v[1] = elliptic_e_sin(0.8, 0.45); v[2] = elliptic_pi_sin(0.8, -0.6, 0.45);
        v[3] = creal(z); v[4] = cimag(z); v[5] = creal(w); v[6] = cimag(w);
        line("legendre", v, 7);
    }

    /* black body */
    {
        double E[5] = { 0.1, 0.5, 1.0, 3.0, 9.0 }, Iv[5] = { -1, -1, -1, -1, -1 }, v[2];
        blackbody(2.5e6, 1.7, 0.4, E, Iv, 5);
        line("blackbody", Iv, 5);
        blackbody(0.0, 1.7, 0.4, E, Iv, 5);                 /* T <= 0: untouched */
        line("blackbody_T0", Iv, 5);
        v[0] = blackbody_photons(2.5e6, 1.7, 0.4, 1.0); v[1] = blackbody_photons_total(2.5e6, 1.7);
        line("photons", v, 2);
    }

    /* helpers answered by the shim itself */
    {
        double x = 1.00005, v[4];
        sim5complex z1 = 1.0 + 2.0 * I, z2 = -3.0, z3 = 1.0 - 2.0 * I, z4 = 0.5;
        int nr = 0;
        v[0] = ensure_range(&x, -1.0, 1.0, 1e-4); v[1] = x;
        sort_roots(&nr, &z1, &z2, &z3, &z4);
        v[2] = nr; v[3] = creal(z1);
        line("helpers", v, 4);
    }
    return 0;
}
