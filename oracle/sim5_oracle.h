/*
 * sim5_oracle.h -- CPU restatement of the SIM5 per-ray hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (sim5_amd/, include/) may include,
 * link or call this file; only tests/, __graft_entry__.smoke() and the cpu_baseline leg
 * of bench.py use it, and only as the checker.
 *
 * Parity status: PINNED.  Every function below is checked bit-for-bit against the
 * unmodified reference compiled by oracle/Makefile into oracle/_ref/libsim5ref.so
 * (tests/test_oracle_vs_ref.py, runs when _ref is present) and against the golden
 * vectors under tests/golden/ that oracle/gen_golden.py captured from that build.
 *
 * Struct layouts are those of the reference headers (same member order and types) so
 * that a record dumped by one side can be compared byte-wise with the other:
 *   orc_geodesic      <-> struct geodesic       reference src/sim5kerr-geod.h:42-68 (240 B)
 *   orc_metric        <-> struct sim5metric     reference src/sim5kerr.h:18-25      ( 64 B)
 *   orc_tetrad        <-> struct sim5tetrad     reference src/sim5kerr.h:27-31      (192 B)
 *   orc_raytrace_data <-> struct raytrace_data  reference src/sim5raytrace.h:26-43  (144 B)
 */
#ifndef SIM5_ORACLE_H
#define SIM5_ORACLE_H

#include <complex.h>

typedef double _Complex orc_cplx;

/* geodesic classes and status codes: reference src/sim5kerr-geod.h:19-37 */
enum { ORC_RR = 40, ORC_RR_DBL = 41, ORC_RR_BH = 42, ORC_RC = 2, ORC_CC = 0 };
enum {
    ORC_OK = 0, ORC_E_Q_ZERO = 1, ORC_E_BOUND = 2, ORC_E_UNKNOWN = 3, ORC_E_RR_DOUBLE = 4,
    ORC_E_CC = 5, ORC_E_Q_RANGE = 7, ORC_E_MUPLUS = 8, ORC_E_MU0 = 9, ORC_E_MM = 10,
    ORC_E_INCL = 11, ORC_E_SPIN = 12
};

typedef struct orc_geodesic {
    double a, alpha, beta, incl, cos_i;
    double l, q;
    orc_cplx r1, r2, r3, r4;
    int nrr, type;
    double m2p, m2m, mm, mK;
    double rp, dmdp_inf;
    double Rpc, Tpp, Tip;
    double k[4];
    double p;
} orc_geodesic;

typedef struct orc_metric { double a, r, m, g00, g11, g22, g33, g03; } orc_metric;
typedef struct orc_tetrad { double e[4][4]; orc_metric metric; } orc_tetrad;

typedef struct orc_raytrace_data {
    int opt_gr, opt_pol;
    double step_epsilon;
    double bh_spin, E, Q;
    orc_cplx WP;
    int pass, refines;
    double dk[4], df[4];
    double kt;
    float error;
} orc_raytrace_data;

typedef struct orc_stokes { double i, q, u, v, tau; } orc_stokes;

/* Novikov-Thorne disk state (the reference keeps these as file statics of type float,
 * src/sim5disk-nt.c:27-32; the oracle keeps them in an explicit object) */
typedef struct orc_disk_nt { float mass, spin, mdot, rms, alpha; int options; } orc_disk_nt;

/* --- Carlson / Legendre / Jacobi (reference src/sim5elliptic.c) ------------------ */
double orc_rf(double x, double y, double z);
double orc_rd(double x, double y, double z);
double orc_rc(double x, double y);
double orc_rj(double x, double y, double z, double p);
double orc_elliptic_k(double m);
double orc_elliptic_f_sin(double sin_phi, double m);
double orc_jacobi_isn(double z, double m);
double orc_jacobi_icn(double z, double m);
double orc_jacobi_itn(double z, double m);
void   orc_jacobi_sncndn(double u, double m, double *sn, double *cn, double *dn);
double orc_jacobi_sn(double u, double m);
double orc_jacobi_cn(double u, double m);
double orc_jacobi_dn(double u, double m);

/* Legendre integrals through Carlson's and the Byrd & Friedman integrals on top (ref: src/sim5elliptic.c:255-1161) */
double orc_elliptic_f_cos(double c, double m);
double orc_elliptic_e_cos(double c, double m);
double orc_elliptic_pi_complete(double n, double m);
double orc_elliptic_pi_cos(double c, double n, double m);
double orc_integral_C2(double u, double m);
double orc_integral_C2_cos(double cn_u, double m);
double orc_integral_Z1(double a, double b, double u, double m);
double orc_integral_Z2(double a, double b, double u, double m);
double orc_integral_Rm1(double a, double u, double m);
double orc_integral_Rm2(double a, double u, double m);
double orc_integral_R1(double a, double u, double m);
double orc_integral_R2(double a, double u, double m);
double orc_integral_R_r0_re(double a, double b, double c, double d, double X);
double orc_integral_R_r0_re_inf(double a, double b, double c, double d);
double orc_integral_R_r0_cc(double a, double b, orc_cplx c, double X);
double orc_integral_R_r0_cc_inf(double a, double b, orc_cplx c);
double orc_integral_R_r1_re(double a, double b, double c, double d, double X);
double orc_integral_R_r1_cc(double a, double b, orc_cplx c, double X1, double X2);
double orc_integral_R_r2_re(double a, double b, double c, double d, double X);
double orc_integral_R_r2_cc(double a, double b, orc_cplx c, double X1, double X2);
double orc_integral_R_rp_re(double a, double b, double c, double d, double p, double X);
double orc_integral_R_rp_re_inf(double a, double b, double c, double d, double p);
double orc_integral_R_rp_cc2(double a, double b, orc_cplx c, double p, double X1, double X2);
double orc_integral_R_rp_cc2_inf(double a, double b, orc_cplx c, double p, double X1);
double orc_integral_T_m0(double a2, double b2, double X);
double orc_integral_T_m2(double a2, double b2, double X);
double orc_integral_T_mp(double a2, double b2, double p, double X);

/* --- Kerr spacetime (reference src/sim5kerr.c) ------------------------------------ */
double orc_r_bh(double a);
double orc_r_ms(double a);
double orc_r_mb(double a);
double orc_r_ph(double a);
void   orc_flat_metric(double r, double m, orc_metric *g);
void   orc_kerr_metric(double a, double r, double m, orc_metric *g);
void   orc_kerr_metric_contravariant(double a, double r, double m, orc_metric *g);
void   orc_kerr_newman_metric(double a, double Q, double r, double m, orc_metric *g);
void   orc_kerr_newman_metric_contravariant(double a, double Q, double r, double m, orc_metric *g);
void   orc_kerr_newman_connection(double a, double Q, double r, double m, double G[4][4][4]);
void   orc_flat_connection(double r, double m, double G[4][4][4]);
void   orc_kerr_connection(double a, double r, double m, double G[4][4][4]);
void   orc_Gamma(double G[4][4][4], double U[4], double V[4], double out[4]);
double orc_dotprod(const double u[4], const double v[4], const orc_metric *g);
void   orc_vector_norm_to(double v[4], double norm, const orc_metric *g);
void   orc_tetrad_zamo(const orc_metric *g, orc_tetrad *t);
void   orc_tetrad_azimuthal(const orc_metric *g, double Omega, orc_tetrad *t);
void   orc_tetrad_surface(const orc_metric *g, double Omega, double V, double dhdr, orc_tetrad *t);
void   orc_bl2on(const double in[4], double out[4], const orc_tetrad *t);
void   orc_on2bl(const double in[4], double out[4], const orc_tetrad *t);
double orc_OmegaK(double r, double a);
double orc_ellK(double r, double a);
double orc_Omega_from_ell(double ell, const orc_metric *g);
double orc_gfactorK(double r, double a, double l);
void   orc_photon_momentum(double a, double r, double m, double l, double q,
                           double r_sign, double m_sign, double k[4]);
void   orc_photon_motion_constants(double a, double r, double m, const double k[4],
                                   double *L, double *Q);
double orc_photon_carter_const(const double k[4], const orc_metric *g);

/* --- elliptic-integral geodesics (reference src/sim5kerr-geod.c) ------------------ */
int    orc_geodesic_init_inf(double i, double a, double alpha, double beta,
                             orc_geodesic *g, int *error);
int    orc_geodesic_init_src(double a, double r, double m, double k[4], int ppc,
                             orc_geodesic *g, int *error);
double orc_geodesic_P_int(const orc_geodesic *g, double r, int ppc);
double orc_geodesic_position_rad(const orc_geodesic *g, double P);
double orc_geodesic_position_pol(const orc_geodesic *g, double P);
double orc_geodesic_dm_sign(const orc_geodesic *g, double P);
void   orc_geodesic_momentum(const orc_geodesic *g, double P, double r, double m, double k[4]);
double orc_geodesic_find_midplane_crossing(const orc_geodesic *g, int order);
/* azimuth and light-travel time (ref: src/sim5kerr-geod.c:463-664) */
double orc_geodesic_position_azm(const orc_geodesic *g, double r, double m, double P);
double orc_geodesic_timedelay(const orc_geodesic *g, double P1, double r1, double m1, double P2, double r2, double m2);
void   orc_geodesic_follow(const orc_geodesic *g, double step, double *P, double *r,
                           double *m, int *status);

/* --- thin disk (reference src/sim5disk-nt.c) --------------------------------------- */
void   orc_disk_nt_setup(orc_disk_nt *d, double M, double a, double mdot, double alpha);
double orc_disk_nt_r_min(const orc_disk_nt *d);
double orc_disk_nt_flux(const orc_disk_nt *d, double r);
double orc_disk_nt_ell(const orc_disk_nt *d, double r);
double orc_disk_nt_mdot(const orc_disk_nt *d);
double orc_disk_nt_sigma(const orc_disk_nt *d, double r);
double orc_disk_nt_lumi(const orc_disk_nt *d);
void   orc_disk_nt_setup_opt(orc_disk_nt *d, double M, double a, double mdot_or_L, double alpha, int options);

/* --- step-wise integrator (reference src/sim5raytrace.c) --------------------------- */
void   orc_raytrace_prepare(double bh_spin, double x[4], double k[4], double precision,
                            int options, orc_raytrace_data *rtd);
void   orc_raytrace(double x[4], double k[4], double *step, orc_raytrace_data *rtd);
double orc_raytrace_error(double x[4], double k[4], orc_raytrace_data *rtd);

/* --- polarization / radiation (reference src/sim5polarization.c, sim5radiation.c) -- */
orc_cplx orc_polarization_constant(const double k[4], const double f[4], const orc_metric *g);
void     orc_polarization_vector(const double k[4], orc_cplx wp, const orc_metric *g, double f[4]);
orc_cplx orc_polarization_constant_infinity(double a, double alpha, double beta, double incl);
double   orc_polarization_angle_rotation(double a, double inc, double alpha, double beta,
                                         orc_cplx kappa);
double   orc_blackbody_Iv(double T, double hardf, double cos_mu, double E);

/* --- whole-pixel recipes used as checkers for the batch kernels -------------------- */
/* class codes written per pixel by orc_disk_pixel (our own bookkeeping of the branches
 * of the caller loop, reference examples/04-disk-image-eqplane/disk-image.c:53-105)  */
enum {
    ORC_PX_ERROR  = 0,  /* geodesic_init_inf failed                       (:66-69) */
    ORC_PX_NAN0   = 1,  /* no first crossing (P is NaN)                   (:74)    */
    ORC_PX_HIT0   = 2,  /* first crossing at r >= rms                     (:83-89) */
    ORC_PX_NAN1   = 3,  /* first crossing inside rms, no second crossing  (:94)    */
    ORC_PX_HIT1   = 4,  /* second crossing at r >= rms                    (:98-103)*/
    ORC_PX_MISS   = 5   /* both crossings inside rms                               */
};
typedef struct orc_pixel {
    int    cls;       /* ORC_PX_*                                   */
    int    gtype;     /* geodesic type (ORC_RR ...) or -1 on error  */
    int    err;       /* GD_* code                                  */
    double r;         /* radius of the accepted crossing (NaN if none) */
    double g;         /* gfactorK                                   */
    double flux;      /* disk_nt_flux(r)  (local flux, not yet * g^4) */
    float  image_f;   /* what disk-image.c stores: (float)(flux*pow(g,4)) */
    float  image_g;   /* (float)g                                   */
} orc_pixel;
void orc_disk_pixel(const orc_disk_nt *d, double inc, double a, double rms,
                    double alpha, double beta, orc_pixel *px);

#endif
