/*
 * sim5lib.h -- the SIM5 scalar C API (host side), served by the MI355X library.
 *
 * Drop-in for the header of the same name in mbursa/sim5 for the null-geodesic hot path: programs
 * written against SIM5 (e.g. its examples/04-disk-image-eqplane/disk-image.c) compile unchanged with
 *      gcc -I<this dir> prog.c <this dir>/sim5lib.c -lm
 * Every function below forwards to the batch entry point of the same name in libsim5gpu.so
 * (include/sim5gpu.h) with n = 1; the library is loaded on first use ($SIM5GPU_LIB, then
 * libsim5gpu.so on the loader path, then the in-tree build).  There is no CPU implementation: if the
 * library or a GPU is missing the first call prints "ERROR: ..." to stderr and exits.
 *
 * A per-ray call costs one kernel launch, so this header is the compatibility layer; programs that
 * trace images should move their pixel loop to sim5gpu_disk_image().
 *
 * Types, names, argument order and the TRUE/FALSE + error-code conventions follow the reference
 * headers (ref: src/sim5kerr-geod.h:19-84, src/sim5kerr.h:18-175, src/sim5raytrace.h:21-53,
 * src/sim5disk-nt.h:17-34, src/sim5polarization.h:19-23, src/sim5radiation.h:19-36,
 * src/sim5elliptic.h:20-56, src/sim5math.h:36-65, src/sim5const.h:24-95).
 */
#ifndef SIM5LIB_H_MI355X
#define SIM5LIB_H_MI355X

#include <complex.h>
#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DEVICEFUNC
#define HOSTFUNC
#define INLINE

#ifndef TRUE
#define TRUE 1
#define FALSE 0
#endif
#define TINY 1e-40

/* ---- math helpers (macros of the reference's sim5math.h) ---- */
#define PI      3.14159265359
#define PI2     6.28318530718
#define PI4     12.5663706144
#define PI_half 1.57079632679
#define sqr(a)   ((a) * (a))
#define sqr2(a)  ((a) * (a))
#define sqr3(a)  ((a) * (a) * (a))
#define sqr4(a)  ((a) * (a) * (a) * (a))
#define sqrt3(a) cbrt(a)
#define sqrt4(a) pow(a,0.25)
#define deg2rad(a) ((a)/180.0*M_PI)
#define rad2deg(a) ((a)*180.0/M_PI)
#define odd(a) ((a%2==1)?1:0)
#define sign(a) ((a) >= 0.0 ? (+1.0) : (-1.0))
#define EE(a) pow(10.0,a)
#define ave(a, b, w) ((1.0-(w))*(a) + (w)*(b))
#define logave(a, b, w) (exp((1.0-(w))*log(a) + (w)*log(b)))
#define inrange(a, min, max) (((a)>=(min))&&((a)<=(max)))
#define ComplexI _Complex_I
typedef double _Complex sim5complex;

/* helpers of sim5math.h that move or compare values (ref src/sim5math.c:16-58, 188-223); no ray arithmetic */
long        sim5round(double num);
long int    factorial(long int n);
double      reduce_angle_pi(double phi);
double      reduce_angle_2pi(double phi);
int         ensure_range(double *val, double min, double max, double acc);
sim5complex makeComplex(double r, double i);
sim5complex nullComplex(void);
double      sim5creal(sim5complex a);
double      sim5cimag(sim5complex a);
/* ref src/sim5polyroots.h:26: real roots first (descending), complex after; *s = number of real roots */
void        sort_roots(int *s, sim5complex *z1, sim5complex *z2, sim5complex *z3, sim5complex *z4);

/* ---- physical constants used by the callers on this path (CGS) ---- */
#define grav_radius         1.476716e+05
#define speed_of_light      2.997925e+10
#define speed_of_light2     8.987554e+20
#define boltzmann_k         1.380650e-16
#define sb_sigma            5.670400e-05
#define parsec              3.085680e+18
#define solar_mass          1.988920e+33
#define grav_const          6.673000e-08
#define planck_h            6.626069e-27
#define Mdot_Edd            2.225475942e+18
#define L_Edd               1.257142540e+38
#define kev2freq            2.417990e+17
#define freq2kev            4.135667e-18
#define kev2erg             1.602177e-09
#define erg2kev             6.241507e+08

/* ---- geodesics ---- */
#define GEOD_TYPE_RR               40
#define GEOD_TYPE_RR_DBL           41
#define GEOD_TYPE_RR_BH            42
#define GEOD_TYPE_RC                2
#define GEOD_TYPE_CC                0

#define GD_OK                           0
#define GD_ERROR_Q_ZERO                 1
#define GD_ERROR_BOUND_GEODESIC         2
#define GD_ERROR_UNKNOWN_SOLUTION       3
#define GD_ERROR_TYPE_RR_DOUBLE         4
#define GD_ERROR_TYPE_CC                5
#define GD_ERROR_Q_RANGE                7
#define GD_ERROR_MUPLUS_RANGE           8
#define GD_ERROR_MU0_RANGE              9
#define GD_ERROR_MM_RANGE              10
#define GD_ERROR_INCL_RANGE            11
#define GD_ERROR_SPIN_RANGE            12

typedef struct geodesic {
    double a, alpha, beta, incl, cos_i;
    double l, q;
    sim5complex r1, r2, r3, r4;
    int    nrr, type;
    double m2p, m2m, mm, mK;
    double rp, dmdp_inf;
    double Rpc, Tpp, Tip;
    double k[4];
    double p;
} geodesic;

int    geodesic_init_inf(double i, double a, double alpha, double beta, geodesic *g, int *error);
int    geodesic_init_src(double a, double r, double m, double k[4], int ppc, geodesic *g, int *error);
double geodesic_P_int(geodesic *g, double r, int ppc);
void   geodesic_position(geodesic *g, double P, double x[4]);      /* an empty stub in the reference too (ref src/sim5kerr-geod.c:266-283) */
double geodesic_position_rad(geodesic *g, double P);
double geodesic_position_pol(geodesic *g, double P);
double geodesic_position_pol_sign_k_theta(geodesic *g, double P);
double geodesic_dm_sign(geodesic *g, double P);
void   geodesic_momentum(geodesic *g, double P, double r, double m, double k[]);
double geodesic_position_azm(geodesic *g, double r, double m, double P);
double geodesic_timedelay(geodesic *g, double P1, double r1, double m1, double P2, double r2, double m2);
double geodesic_find_midplane_crossing(geodesic *g, int order);
void   geodesic_follow(geodesic *g, double step, double *P, double *r, double *m, int *status);

/* ---- Kerr spacetime ---- */
struct sim5metric { double a, r, m; double g00, g11, g22, g33, g03; };
typedef struct sim5metric sim5metric;
struct sim5tetrad { double e[4][4]; sim5metric metric; };
typedef struct sim5tetrad sim5tetrad;

void   flat_metric(double r, double m, sim5metric *metric);
void   flat_metric_contravariant(double r, double m, sim5metric *metric);
void   kerr_metric(double a, double r, double m, sim5metric *metric);
void   kerr_metric_contravariant(double a, double r, double m, sim5metric *metric);
void   kerr_newman_metric(double a, double Q, double r, double m, sim5metric *metric);
void   kerr_newman_metric_contravariant(double a, double Q, double r, double m, sim5metric *metric);
void   kerr_newman_connection(double a, double Q, double r, double m, double G[4][4][4]);
void   flat_connection(double r, double m, double G[4][4][4]);
void   kerr_connection(double a, double r, double m, double G[4][4][4]);
void   Gamma(double G[4][4][4], double U[4], double V[4], double result[4]);
void   vector_set(double x[4], double x0, double x1, double x2, double x3);
void   vector_copy(double src[4], double dst[4]);
void   vector_covariant(double V1[4], double V2[4], sim5metric *m);
double vector_norm(double V[4], sim5metric *m);
double vector_3norm(double V[4]);
void   vector_multiply(double V[4], double factor);
double dotprod(double V1[4], double V2[4], sim5metric *m);
void   vector_norm_to(double V[4], double norm, sim5metric *m);
void   vector_norm_to_null(double V[4], double V0, sim5metric *m);
void   tetrad_general(sim5metric *m, double U[], sim5tetrad *t);
void   tetrad_radial(sim5metric *m, double v_r, sim5tetrad *t);
void   tetrad_zamo(sim5metric *m, sim5tetrad *t);
void   tetrad_azimuthal(sim5metric *m, double Omega, sim5tetrad *t);
void   tetrad_surface(sim5metric *m, double Omega, double V, double dhdr, sim5tetrad *t);
void   bl2on(double Vin[4], double Vout[4], sim5tetrad *t);
void   on2bl(double Vin[4], double Vout[4], sim5tetrad *t);
double r_bh(double a);
double r_ms(double a);
double r_mb(double a);
double r_ph(double a);
double OmegaK(double r, double a);
double ellK(double r, double a);
double omega_r(double r, double a);
double omega_z(double r, double a);
double Omega_from_ell(double ell, sim5metric *m);
double ell_from_Omega(double Omega, sim5metric *m);
double gfactorK(double r, double a, double l);
void   photon_momentum(double a, double r, double m, double l, double q, double r_sign, double m_sign, double k[4]);
void   photon_motion_constants(double a, double r, double m, double k[4], double *L, double *Q);
double photon_carter_const(double k[4], sim5metric *metric);
void   fourvelocity_zamo(sim5metric *m, double U[4]);
void   fourvelocity_azimuthal(double Omega, sim5metric *m, double U[4]);
void   fourvelocity_radial(double vr, sim5metric *m, double U[4]);
double fourvelocity_norm(double U1, double U2, double U3, sim5metric *m);
void   fourvelocity(double U1, double U2, double U3, sim5metric *m, double U[]);

/* ---- step-wise integrator ---- */
#define RTOPT_NONE              0
#define RTOPT_FLAT              1
#define RTOPT_POLARIZATION      2

typedef struct raytrace_data {
    int opt_gr, opt_pol;
    double step_epsilon;
    double bh_spin, E, Q;
    sim5complex WP;
    int pass, refines;
    double dk[4], df[4];
    double kt;
    float error;
} raytrace_data;

void   raytrace_prepare(double bh_spin, double x[4], double k[4], double presision_factor, int options, raytrace_data *rtd);
void   raytrace(double x[4], double k[4], double *step, raytrace_data *rtd);
double raytrace_error(double x[4], double k[4], raytrace_data *rtd);

/* ---- thin disk ---- */
#define DISK_NT_OPTION_LUMINOSITY     1
int    disk_nt_setup(double M, double a, double mdot_or_L, double alpha, int options);
void   disk_nt_done(void);
double disk_nt_r_min(void);
double disk_nt_flux(double r);
double disk_nt_lumi(void);
double disk_nt_mdot(void);
double disk_nt_sigma(double r);
double disk_nt_ell(double r);
double disk_nt_vr(double r);
double disk_nt_h(double r);
double disk_nt_dhdr(double r);
void   disk_nt_dump(char *filename);

/* ---- polarization, radiation ---- */
typedef struct stokes_params { double i, q, u, v, tau; } stokes_params;
static const stokes_params stokes_null = {0.0, 0.0, 0.0, 0.0, 0.0};

sim5complex polarization_constant(double k[4], double f[4], sim5metric *metric);
void        polarization_vector(double k[4], sim5complex wp, sim5metric *metric, double f[4]);
sim5complex polarization_constant_infinity(double a, double alpha, double beta, double incl);
double      polarization_angle_rotation(double a, double inc, double alpha, double beta, sim5complex kappa);
double      blackbody_Iv(double T, double hardf, double cos_mu, double E);
void        blackbody(double T, double hardf, double cos_mu, double E[], double Iv[], int en_bins);
double      blackbody_photons(double T, double hardf, double cos_mu, double E);
double      blackbody_photons_total(double T, double hardf);

/* ---- elliptic functions ---- */
double rf(double x, double y, double z);
double rd(double x, double y, double z);
double rc(double x, double y);
double rj(double x, double y, double z, double p);
double elliptic_k(double m);
double jacobi_isn(double z, double m);
double jacobi_icn(double z, double m);
double jacobi_itn(double z, double m);
double jacobi_sn(double u, double m);
double jacobi_cn(double u, double m);
double jacobi_dn(double u, double m);
void   jacobi_sncndn(double u, double m, double *sn, double *cn, double *dn);
/* Legendre integrals of the 1st-3rd kind and the radial / polar integrals built on them
   (ref: src/sim5elliptic.h:25-56); a `sim5complex c` is the complex root u + i v */
double elliptic_f(double phi, double m);
double elliptic_f_sin(double sin_phi, double m);
double elliptic_f_cos(double cos_phi, double m);
double elliptic_e_sin(double sin_phi, double m);
double elliptic_e_cos(double cos_phi, double m);
double elliptic_pi_complete(double n, double m);
double elliptic_pi_cos(double cos_phi, double n, double m);
double elliptic_pi_sin(double sin_phi, double n, double m);
sim5complex elliptic_pi(double phi, double n, double m);
double integral_R_r0_re(double a, double b, double c, double d, double X);
double integral_R_r0_cc(double a, double b, sim5complex c, double X);
double integral_R_r0_re_inf(double a, double b, double c, double d);
double integral_R_r0_cc_inf(double a, double b, sim5complex c);
double integral_R_r1_re(double a, double b, double c, double d, double X);
double integral_R_r1_cc(double a, double b, sim5complex c, double X1, double X2);
double integral_R_r2_re(double a, double b, double c, double d, double X);
double integral_R_r2_cc(double a, double b, sim5complex c, double X1, double X2);
double integral_R_rp_re(double a, double b, double c, double d, double p, double X);
double integral_R_rp_cc2(double a, double b, sim5complex c, double p, double X1, double X2);
double integral_R_rp_re_inf(double a, double b, double c, double d, double p);
double integral_R_rp_cc2_inf(double a, double b, sim5complex c, double p, double X1);
double integral_T_m0(double a2, double b2, double X);
double integral_T_m2(double a2, double b2, double X);
double integral_T_mp(double a2, double b2, double p, double X);

#ifdef __cplusplus
}
#endif
#endif
