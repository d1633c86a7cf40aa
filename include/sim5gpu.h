/*
 * sim5gpu.h -- C-ABI of the MI355X (gfx950) Kerr ray tracer.
 *
 * Plain C: pointers, sizes and POD structs only; no HIP, torch or C++ types cross this
 * boundary.  Implemented by sim5_amd/lib/libsim5gpu.so (hand-written HIP kernels).
 * There is NO CPU fallback behind these entry points: without a usable GPU every
 * compute call returns SIM5GPU_E_NO_DEVICE / SIM5GPU_E_HIP and writes nothing.
 *
 * Two groups of entry points:
 *
 *  (1) Batch forms of the SIM5 per-ray functions ("sim5gpu_<sim5 name>").  Each mirrors
 *      the argument meaning, output struct layout and error behaviour of the SIM5
 *      function it replaces, for n independent rays held in caller-owned HOST arrays.
 *      A reference-side binding (cgo/ctypes/plain C) calls these in place of its
 *      per-ray loop; sim5_amd/host/sim5lib.c implements the unchanged SIM5 scalar API
 *      on top of them (n = 1).
 *
 *  (2) Whole-job kernels working on DEVICE buffers (pointers from hipMalloc, from
 *      sim5gpu_malloc below, or from torch tensors' data_ptr()), asynchronous on a
 *      caller-supplied stream: the thin-disk image loop, the polarized image and the
 *      step-wise (Verlet) ray tracer with radiative transfer.
 *
 * Citations "ref:" are file:line in the reference tree (mbursa/sim5).
 */
#ifndef SIM5GPU_H
#define SIM5GPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes ------------------------------------------------------------- */
#define SIM5GPU_OK            0
#define SIM5GPU_E_NO_DEVICE  -1   /* no HIP device visible                          */
#define SIM5GPU_E_HIP        -2   /* a HIP runtime call failed (see last_error)     */
#define SIM5GPU_E_ARG        -3   /* invalid argument (NULL pointer, bad size, ...)  */
#define SIM5GPU_E_NOT_SETUP  -4   /* disk model used before sim5gpu_disk_nt_setup   */

/* ---- geodesic classes and per-ray status: ref src/sim5kerr-geod.h:19-37 -------- */
#define SIM5GPU_GEOD_TYPE_RR      40
#define SIM5GPU_GEOD_TYPE_RR_DBL  41
#define SIM5GPU_GEOD_TYPE_RR_BH   42
#define SIM5GPU_GEOD_TYPE_RC       2
#define SIM5GPU_GEOD_TYPE_CC       0

#define SIM5GPU_GD_OK                      0
#define SIM5GPU_GD_ERROR_Q_ZERO            1
#define SIM5GPU_GD_ERROR_BOUND_GEODESIC    2
#define SIM5GPU_GD_ERROR_UNKNOWN_SOLUTION  3
#define SIM5GPU_GD_ERROR_TYPE_RR_DOUBLE    4
#define SIM5GPU_GD_ERROR_TYPE_CC           5
#define SIM5GPU_GD_ERROR_Q_RANGE           7
#define SIM5GPU_GD_ERROR_MUPLUS_RANGE      8
#define SIM5GPU_GD_ERROR_MU0_RANGE         9
#define SIM5GPU_GD_ERROR_MM_RANGE         10
#define SIM5GPU_GD_ERROR_INCL_RANGE       11
#define SIM5GPU_GD_ERROR_SPIN_RANGE       12

/* ---- options of the step-wise integrator: ref src/sim5raytrace.h:21-23 --------- */
#define SIM5GPU_RTOPT_NONE          0
#define SIM5GPU_RTOPT_FLAT          1
#define SIM5GPU_RTOPT_POLARIZATION  2

/* ---- POD records with the reference layouts ------------------------------------- */

/* 240 B; byte-compatible with `struct geodesic` (ref src/sim5kerr-geod.h:42-68).
 * The four roots are C99 `double _Complex` there; {re, im} pairs here. */
typedef struct sim5gpu_geodesic {
    double a, alpha, beta, incl, cos_i;
    double l, q;
    double r1[2], r2[2], r3[2], r4[2];
    int    nrr, type;
    double m2p, m2m, mm, mK;
    double rp, dmdp_inf;
    double Rpc, Tpp, Tip;
    double k[4];
    double p;
} sim5gpu_geodesic;

/* 64 B; `struct sim5metric` (ref src/sim5kerr.h:18-25) */
typedef struct sim5gpu_metric { double a, r, m, g00, g11, g22, g33, g03; } sim5gpu_metric;

/* 192 B; `struct sim5tetrad` (ref src/sim5kerr.h:27-31) */
typedef struct sim5gpu_tetrad { double e[4][4]; sim5gpu_metric metric; } sim5gpu_tetrad;

/* 144 B; `struct raytrace_data` (ref src/sim5raytrace.h:26-43) */
typedef struct sim5gpu_raytrace_data {
    int    opt_gr, opt_pol;
    double step_epsilon;
    double bh_spin, E, Q;
    double WP[2];
    int    pass, refines;
    double dk[4], df[4];
    double kt;
    float  error;
} sim5gpu_raytrace_data;

/* 40 B; `struct stokes_params` (ref src/sim5radiation.h:19-25) */
typedef struct sim5gpu_stokes { double i, q, u, v, tau; } sim5gpu_stokes;

/* ---- runtime ---------------------------------------------------------------------- */
int         sim5gpu_device_count(void);            /* 0 when no GPU is visible       */
int         sim5gpu_set_device(int device);
const char *sim5gpu_last_error(void);              /* text of the last failure        */
const char *sim5gpu_version(void);
int         sim5gpu_malloc(void **dptr, size_t bytes);
int         sim5gpu_free(void *dptr);
int         sim5gpu_memcpy_h2d(void *dst, const void *src, size_t bytes);
int         sim5gpu_memcpy_d2h(void *dst, const void *src, size_t bytes);
int         sim5gpu_memset(void *dst, int value, size_t bytes);
int         sim5gpu_synchronize(void *stream);     /* stream == NULL: default stream  */
/* HIP events on a caller stream, for timing launches where they are issued (bench.py) */
int         sim5gpu_event_create(void **event);
int         sim5gpu_event_record(void *event, void *stream);
int         sim5gpu_event_elapsed_ms(void *start, void *stop, float *ms);  /* waits for `stop` */
int         sim5gpu_event_destroy(void *event);

/* ==================================================================================== */
/* (1) batch forms of the SIM5 per-ray API (HOST arrays, synchronous)                   */
/* ==================================================================================== */

/* geodesic_init_inf (ref src/sim5kerr-geod.c:42-100) for n rays.
 * ok[i] = TRUE/FALSE return value, error[i] = GD_* code (either may be NULL). */
int sim5gpu_geodesic_init_inf(size_t n, const double *incl, const double *a,
                              const double *alpha, const double *beta,
                              sim5gpu_geodesic *g, int *error, int *ok);

/* geodesic_init_inf plus the values the caller loop of ref examples/04-disk-image-eqplane/disk-image.c:62-100 asks for
 * next -- one launch and one round trip per ray instead of five.  For the orders k = 0, 1:
 *   P[k] = geodesic_find_midplane_crossing(g, k);  r[k] = geodesic_position_rad(g, P[k]) if P[k] is a number (have_r[k]);
 *   g[k] = gfactorK(r[k], a, g->l) and flux[k] = disk_nt_flux(r[k]) if r[k] is a number (flux only when the disk model has
 *   been set up: flux_valid).  Every value comes from the device routine of the single entry point, same arguments. */
typedef struct sim5gpu_geodesic_chain {
    double P[2], r[2], g[2], flux[2];
    double a, l;               /* the arguments g[] was computed with: the caller's spin and the geodesic's l          */
    int    have_r[2];          /* position_rad was evaluated for this order                                           */
    int    valid;              /* geodesic_init_inf returned TRUE                                                     */
    int    flux_valid;         /* sim5gpu_disk_nt_setup had been called: flux[] holds disk_nt_flux of the model then set */
} sim5gpu_geodesic_chain;
int sim5gpu_geodesic_init_inf_chain(size_t n, const double *incl, const double *a, const double *alpha, const double *beta,
                                    sim5gpu_geodesic *g, int *error, int *ok, sim5gpu_geodesic_chain *chain);
/* the same record in the library's FAST arithmetic (the whole-image kernels' default: sim5_amd/csrc/k_chain.hip): agrees with
 * the entry point above to ~1e-12 relative, in a third of its latency -- which is what a caller of the scalar API waits for */
int sim5gpu_geodesic_init_inf_chain_fast(size_t n, const double *incl, const double *a, const double *alpha, const double *beta,
                                         sim5gpu_geodesic *g, int *error, int *ok, sim5gpu_geodesic_chain *chain);

/* geodesic_init_src (ref src/sim5kerr-geod.c:106-173); k is n x 4 */
int sim5gpu_geodesic_init_src(size_t n, const double *a, const double *r, const double *m,
                              const double *k, const int *ppc,
                              sim5gpu_geodesic *g, int *error, int *ok);

/* geodesic_find_midplane_crossing (ref src/sim5kerr-geod.c:846-885); NaN = no crossing */
int sim5gpu_geodesic_find_midplane_crossing(size_t n, const sim5gpu_geodesic *g,
                                            const int *order, double *P);

/* geodesic_P_int (ref src/sim5kerr-geod.c:179-263) */
int sim5gpu_geodesic_P_int(size_t n, const sim5gpu_geodesic *g, const double *r,
                           const int *ppc, double *P);

/* geodesic_position_rad / _pol (ref src/sim5kerr-geod.c:291-357, 363-407) */
int sim5gpu_geodesic_position_rad(size_t n, const sim5gpu_geodesic *g, const double *P, double *r);
int sim5gpu_geodesic_position_pol(size_t n, const sim5gpu_geodesic *g, const double *P, double *m);

/* geodesic_dm_sign (ref src/sim5kerr-geod.c:737-781) */
int sim5gpu_geodesic_dm_sign(size_t n, const sim5gpu_geodesic *g, const double *P, double *sign);

/* geodesic_momentum (ref src/sim5kerr-geod.c:787-840); r = m = 0 means "derive from P";
 * k is n x 4 */
int sim5gpu_geodesic_momentum(size_t n, const sim5gpu_geodesic *g, const double *P,
                              const double *r, const double *m, double *k);

/* geodesic_follow (ref src/sim5kerr-geod.c:891-925); P, r, m are in/out */
int sim5gpu_geodesic_follow(size_t n, const sim5gpu_geodesic *g, const double *step,
                            double *P, double *r, double *m, int *status);

/* photon_momentum (ref src/sim5kerr.c:1151-1213); k is n x 4 */
int sim5gpu_photon_momentum(size_t n, const double *a, const double *r, const double *m,
                            const double *l, const double *q,
                            const double *r_sign, const double *m_sign, double *k);

/* photon_motion_constants / photon_carter_const (ref src/sim5kerr.c:1217-1269) */
int sim5gpu_photon_motion_constants(size_t n, const double *a, const double *r, const double *m,
                                    const double *k, double *L, double *Q);
int sim5gpu_photon_carter_const(size_t n, const double *k, const sim5gpu_metric *metric, double *Q);

/* r_mb, r_ph: marginally bound and photon orbit (ref src/sim5kerr.c:1007-1034; used by examples/01-kerr-spacetime) */
int sim5gpu_r_mb(size_t n, const double *a, double *r);
int sim5gpu_r_ph(size_t n, const double *a, double *r);

/* r_bh, r_ms, OmegaK, ellK, Omega_from_ell, dotprod (ref src/sim5kerr.c:981, 994, 1037, 1050, 1101, 609);
 * dotprod: metric == NULL means the Minkowski product -v0w0 + v1w1 + v2w2 + v3w3, as in SIM5 */
int sim5gpu_r_bh(size_t n, const double *a, double *r);
int sim5gpu_r_ms(size_t n, const double *a, double *r);
int sim5gpu_OmegaK(size_t n, const double *r, const double *a, double *Omega);
int sim5gpu_ellK(size_t n, const double *r, const double *a, double *ell);
int sim5gpu_Omega_from_ell(size_t n, const double *ell, const sim5gpu_metric *metric, double *Omega);
int sim5gpu_dotprod(size_t n, const double *v1, const double *v2, const sim5gpu_metric *metric, double *out);

/* gfactorK (ref src/sim5kerr.c:1128-1141) */
int sim5gpu_gfactorK(size_t n, const double *r, const double *a, const double *l, double *g);

/* kerr_metric / kerr_connection (ref src/sim5kerr.c:75-101, 233-316); G is n x 64 */
int sim5gpu_kerr_metric(size_t n, const double *a, const double *r, const double *m,
                        sim5gpu_metric *metric);
int sim5gpu_kerr_connection(size_t n, const double *a, const double *r, const double *m, double *G);

/* tetrad_zamo / tetrad_azimuthal / tetrad_surface (ref src/sim5kerr.c:678, 766, 818) */
int sim5gpu_tetrad_zamo(size_t n, const sim5gpu_metric *metric, sim5gpu_tetrad *t);
int sim5gpu_tetrad_azimuthal(size_t n, const sim5gpu_metric *metric, const double *Omega,
                             sim5gpu_tetrad *t);
int sim5gpu_tetrad_surface(size_t n, const sim5gpu_metric *metric, const double *Omega,
                           const double *V, const double *dhdr, sim5gpu_tetrad *t);

/* vector_norm_to (ref src/sim5kerr.c:553-573): scales each vector (n x 4, in place) to V.V = norm; metric may be
 * NULL (Minkowski), as in the reference */
int sim5gpu_vector_norm_to(size_t n, double *v, const double *norm, const sim5gpu_metric *metric);

/* bl2on / on2bl (ref src/sim5kerr.c:926-970); vectors are n x 4 */
int sim5gpu_bl2on(size_t n, const double *vin, double *vout, const sim5gpu_tetrad *t);
int sim5gpu_on2bl(size_t n, const double *vin, double *vout, const sim5gpu_tetrad *t);

/* Carlson R_F and the Jacobi inverses built on it (ref src/sim5elliptic.c:19-52, 218-225,
 * 481-528, 536-598).  which: 0 = rf(x,y,z), 1 = elliptic_k(x), 2 = jacobi_isn(x,y),
 * 3 = jacobi_icn(x,y), 4 = jacobi_itn(x,y), 5 = jacobi_sn(x,y), 6 = jacobi_cn(x,y),
 * 7 = jacobi_dn(x,y), 8 = rd(x,y,z), 9 = rc(x,y), 10 = rj(x,y,z,w), 11 = elliptic_f_sin(x,y) */
int sim5gpu_elliptic(int which, size_t n, const double *x, const double *y, const double *z,
                     const double *w, double *out);

/* geodesic_position_azm: change of azimuth between infinity and the point (r, m) at position integral P
 * (ref src/sim5kerr-geod.c:463-556); NaN for RR_DBL / RR_BH / CC geodesics as in the reference.
 * geodesic_timedelay: light-travel time between two points of a geodesic (ref :560-664; the reference
 * evaluates the radial part, its polar part is commented out); r = 0 asks for r, m to be derived from P. */
int sim5gpu_geodesic_position_azm(size_t n, const sim5gpu_geodesic *g, const double *r, const double *m,
                                  const double *P, double *phi);
int sim5gpu_geodesic_timedelay(size_t n, const sim5gpu_geodesic *g, const double *P1, const double *r1,
                               const double *m1, const double *P2, const double *r2, const double *m2,
                               double *dt);

/* The integrals under them (ref src/sim5elliptic.c:255-1161).  args holds nargs rows of n values, one row
 * per argument in the reference's order; a `sim5complex c` argument is two rows (re, im).  which:
 *  0 elliptic_f_cos(c,m)        1 elliptic_e_cos(c,m)       2 elliptic_pi_complete(n,m)  3 elliptic_pi_cos(c,n,m)
 *  4 integral_C2(u,m)           5 integral_C2_cos(cn,m)     6 integral_Z1(a,b,u,m)       7 integral_Z2(a,b,u,m)
 *  8 integral_Rm1(a,u,m)        9 integral_Rm2(a,u,m)      10 integral_R1(a,u,m)        11 integral_R2(a,u,m)
 * 12 integral_R_r0_re(a,b,c,d,X)         13 integral_R_r0_re_inf(a,b,c,d)      14 integral_R_r1_re(a,b,c,d,X)
 * 15 integral_R_r2_re(a,b,c,d,X)         16 integral_R_rp_re(a,b,c,d,p,X)      17 integral_R_rp_re_inf(a,b,c,d,p)
 * 18 integral_R_r0_cc(a,b,c,X)           19 integral_R_r0_cc_inf(a,b,c)        20 integral_R_r1_cc(a,b,c,X1,X2)
 * 21 integral_R_r2_cc(a,b,c,X1,X2)       22 integral_R_rp_cc2(a,b,c,p,X1,X2)   23 integral_R_rp_cc2_inf(a,b,c,p,X1)
 * 24 integral_T_m0(a2,b2,X)              25 integral_T_m2(a2,b2,X)             26 integral_T_mp(a2,b2,p,X) */
int sim5gpu_integral(int which, size_t n, int nargs, const double *args, double *out);

/* Novikov-Thorne disk (ref src/sim5disk-nt.c:37-266, 371-385).  As in SIM5 the disk model is process-global state
 * set once by disk_nt_setup.  options: 0 = mdot_or_L is the accretion rate; SIM5GPU_DISK_NT_OPTION_LUMINOSITY
 * (= DISK_NT_OPTION_LUMINOSITY, ref src/sim5disk-nt.h:17) = it is the luminosity and the accretion rate is found by
 * the reference's bisection over disk_nt_lumi().  disk_nt_lumi is the reference's Simpson integral of the flux
 * (ref :151-188), its integrand evaluated on the device; disk_nt_sigma the column density (ref :204-250). */
#define SIM5GPU_DISK_NT_OPTION_LUMINOSITY 1
int sim5gpu_disk_nt_setup(double M, double a, double mdot_or_L, double alpha, int options);
/* Number of successful sim5gpu_disk_nt_setup calls of this process so far.  The model is process-global (ref
 * src/sim5disk-nt.c:27-32), so anything that remembers a disk_nt_* value -- the host shim's per-ray and look-ahead records
 * (sim5_amd/host/sim5lib.c) -- stamps it with this counter and compares before it answers from memory, whoever made the
 * set-up call (C shim, ctypes, another library in the process). */
unsigned long sim5gpu_disk_nt_generation(void);
int sim5gpu_disk_nt_r_min(double *r_min);
int sim5gpu_disk_nt_mdot(double *mdot);
int sim5gpu_disk_nt_lumi(double *lumi);
int sim5gpu_disk_nt_flux(size_t n, const double *r, double *flux);
int sim5gpu_disk_nt_sigma(size_t n, const double *r, double *sigma);
int sim5gpu_disk_nt_ell(size_t n, const double *r, double *ell);

/* raytrace_prepare / raytrace / raytrace_error (ref src/sim5raytrace.c:44-94, 109-245,
 * 328-343).  x, k are n x 4 and in/out; step is in/out.  sim5gpu_raytrace makes `nsteps`
 * consecutive calls of raytrace() per ray with the same *step cap re-applied each call
 * (nsteps = 1 is exactly one SIM5 raytrace() call). */
int sim5gpu_raytrace_prepare(size_t n, const double *bh_spin, const double *x, const double *k,
                             const double *precision, const int *options,
                             sim5gpu_raytrace_data *rtd);
int sim5gpu_raytrace(size_t n, double *x, double *k, double *step, sim5gpu_raytrace_data *rtd,
                     int nsteps);
int sim5gpu_raytrace_error(size_t n, const double *x, const double *k,
                           const sim5gpu_raytrace_data *rtd, double *err);
/* ONE ray, `nsteps` (1 .. 4096) consecutive raytrace() calls with the cap `step_cap` re-applied at each; records[j] = what
 * the j-th call leaves in x, k, *step and *rtd (bytes of *rtd the integrator does not write are the caller's).  For callers
 * that make the calls one by one -- the loop of ref README.md:184-193 / src/sim5unittests.c:116-127 through the scalar API:
 * one launch per nsteps calls (sim5_amd/host/sim5lib.c serves a call from a record only after a bit-for-bit check of x, k,
 * *step and *rtd against what the record was made from). */
typedef struct sim5gpu_raytrace_step { double x[4], k[4], step; sim5gpu_raytrace_data rtd; } sim5gpu_raytrace_step;   /* 216 B */
int sim5gpu_raytrace_record(const double *x, const double *k, double step_cap, const sim5gpu_raytrace_data *rtd, int nsteps,
                            sim5gpu_raytrace_step *records);

/* polarization_constant / _vector / _constant_infinity / _angle_rotation
 * (ref src/sim5polarization.c:145-158, 14-105, 249-258, 272-285); wp is n x 2 {re, im} */
int sim5gpu_polarization_constant(size_t n, const double *k, const double *f,
                                  const sim5gpu_metric *metric, double *wp);
int sim5gpu_polarization_vector(size_t n, const double *k, const double *wp,
                                const sim5gpu_metric *metric, double *f);
int sim5gpu_polarization_constant_infinity(size_t n, const double *a, const double *alpha,
                                           const double *beta, const double *incl, double *wp);
int sim5gpu_polarization_angle_rotation(size_t n, const double *a, const double *inc,
                                        const double *alpha, const double *beta,
                                        const double *wp, double *angle);

/* blackbody_Iv (ref src/sim5radiation.c:27-49) */
int sim5gpu_blackbody_Iv(size_t n, const double *T, const double *hardf, const double *cos_mu,
                         const double *E, double *Iv);

/* ---- (1b) the remaining public prototypes of the SIM5 headers a caller of this path may link
 * (ref src/sim5kerr.h:36-175, src/sim5kerr-geod.h:77, src/sim5elliptic.h:25-33, src/sim5radiation.h:33-35).
 * Vectors are n x 4, a connection is n x 64 (G[i][j][k] row-major, the reference's double[4][4][4]). ---- */

/* flat_metric / flat_metric_contravariant / kerr_metric_contravariant / flat_connection
 * (ref src/sim5kerr.c:31-50, 54-71, 105-132, 199-229) */
int sim5gpu_flat_metric(size_t n, const double *r, const double *m, sim5gpu_metric *metric);
int sim5gpu_flat_metric_contravariant(size_t n, const double *r, const double *m, sim5gpu_metric *metric);
int sim5gpu_kerr_metric_contravariant(size_t n, const double *a, const double *r, const double *m,
                                      sim5gpu_metric *metric);
/* kerr_newman_metric / kerr_newman_metric_contravariant / kerr_newman_connection (charge Q;
 * ref src/sim5kerr.h:49,52,61, src/sim5kerr.c:136-194, 321-397); G is n x 64 as for kerr_connection */
int sim5gpu_kerr_newman_metric(size_t n, const double *a, const double *Q, const double *r, const double *m,
                               sim5gpu_metric *metric);
int sim5gpu_kerr_newman_metric_contravariant(size_t n, const double *a, const double *Q, const double *r, const double *m,
                                             sim5gpu_metric *metric);
int sim5gpu_kerr_newman_connection(size_t n, const double *a, const double *Q, const double *r, const double *m, double *G);
int sim5gpu_flat_connection(size_t n, const double *r, const double *m, double *G);

/* Gamma: -G^i_(jk) U^j V^k for a connection handed in by the caller (ref src/sim5kerr.c:422-440) */
int sim5gpu_Gamma(size_t n, const double *G, const double *U, const double *V, double *result);

/* vector_covariant / vector_norm / vector_3norm / vector_norm_to_null (ref src/sim5kerr.c:477-532, 577-605);
 * metric == NULL: Minkowski, as in the reference; vector_norm_to_null scales v in place */
int sim5gpu_vector_covariant(size_t n, const double *v1, double *v2, const sim5gpu_metric *metric);
int sim5gpu_vector_norm(size_t n, const double *v, const sim5gpu_metric *metric, double *out);
int sim5gpu_vector_3norm(size_t n, const double *v, double *out);
int sim5gpu_vector_norm_to_null(size_t n, double *v, const double *V0, const sim5gpu_metric *metric);

/* tetrad_general / tetrad_radial (ref src/sim5kerr.c:630-674, 715-762) */
int sim5gpu_tetrad_general(size_t n, const sim5gpu_metric *metric, const double *U, sim5gpu_tetrad *t);
int sim5gpu_tetrad_radial(size_t n, const sim5gpu_metric *metric, const double *v_r, sim5gpu_tetrad *t);

/* omega_r / omega_z: epicyclic frequencies; ell_from_Omega (ref src/sim5kerr.c:1076-1098, 1114-1124) */
int sim5gpu_omega_r(size_t n, const double *r, const double *a, double *out);
int sim5gpu_omega_z(size_t n, const double *r, const double *a, double *out);
int sim5gpu_ell_from_Omega(size_t n, const double *Omega, const sim5gpu_metric *metric, double *ell);

/* fourvelocity_zamo / _azimuthal / _radial / _norm / fourvelocity (ref src/sim5kerr.c:1278-1353) */
int sim5gpu_fourvelocity_zamo(size_t n, const sim5gpu_metric *metric, double *U);
int sim5gpu_fourvelocity_azimuthal(size_t n, const double *Omega, const sim5gpu_metric *metric, double *U);
int sim5gpu_fourvelocity_radial(size_t n, const double *vr, const sim5gpu_metric *metric, double *U);
int sim5gpu_fourvelocity_norm(size_t n, const double *U1, const double *U2, const double *U3,
                              const sim5gpu_metric *metric, double *out);
int sim5gpu_fourvelocity(size_t n, const double *U1, const double *U2, const double *U3,
                         const sim5gpu_metric *metric, double *U);

/* geodesic_position_pol_sign_k_theta (ref src/sim5kerr-geod.c:413-457): +1 / -1, NaN for RR_DBL / RR_BH */
int sim5gpu_geodesic_position_pol_sign_k_theta(size_t n, const sim5gpu_geodesic *g, const double *P, double *sign);

/* Legendre integrals by angle / by sine (ref src/sim5elliptic.c:235-252, 339-357, 453-474, 382-423).  which:
 * 0 = elliptic_f(phi = x, m), 1 = elliptic_e_sin(sin_phi = x, m), 2 = elliptic_pi_sin(sin_phi = x, nn, m),
 * 3 = elliptic_pi(phi = x, nn, m): complex, out is n x 2 {re, im} (the imaginary part is non-zero for nn > 1) */
int sim5gpu_legendre(int which, size_t n, const double *x, const double *nn, const double *m, double *out);

/* blackbody: the spectrum form of blackbody_Iv, Iv[i] for n_energies energies of ONE black body (ref
 * src/sim5radiation.c:53-78; T <= 0 leaves Iv untouched, as there); blackbody_photons / blackbody_photons_total
 * (ref :83-114) */
int sim5gpu_blackbody(double T, double hardf, double cos_mu, size_t n_energies, const double *E, double *Iv);
int sim5gpu_blackbody_photons(size_t n, const double *T, const double *hardf, const double *cos_mu,
                              const double *E, double *out);
int sim5gpu_blackbody_photons_total(size_t n, const double *T, const double *hardf, double *out);

/* ==================================================================================== */
/* (2) whole-job kernels (DEVICE buffers, asynchronous on `stream`)                     */
/* ==================================================================================== */

/* per-pixel outcome of the thin-disk loop; the numbering is this project's own
 * bookkeeping of the branches of ref examples/04-disk-image-eqplane/disk-image.c:53-105 */
#define SIM5GPU_PX_ERROR 0   /* geodesic_init_inf rejected the ray       (:66-69)  */
#define SIM5GPU_PX_NAN0  1   /* no first equatorial crossing              (:74)     */
#define SIM5GPU_PX_HIT0  2   /* first crossing on the disk, r >= r_ms     (:83-89)  */
#define SIM5GPU_PX_NAN1  3   /* first inside r_ms, no second crossing     (:94)     */
#define SIM5GPU_PX_HIT1  4   /* second crossing on the disk               (:98-103) */
#define SIM5GPU_PX_MISS  5   /* both crossings inside r_ms                          */

/* Job description of one thin-disk image (or a row tile of it).  The pixel -> impact
 * parameter map is the one of ref disk-image.c:57-58 for an nx x ny image whose
 * half-width is rmax.  Rows [y0, y1) are traced; outputs are packed row-major arrays of
 * (y1-y0) x nx elements (tile-local), so a row-tile shard is contiguous in the full image. */
typedef struct sim5gpu_image_desc {
    int    nx, ny;          /* full image size in pixels                               */
    int    y0, y1;          /* row range traced by this call, 0 <= y0 < y1 <= ny        */
    double a;               /* black-hole spin                                           */
    double incl;            /* observer inclination [rad]                                */
    double rmax;            /* half-width of the view [GM/c^2]; <= 0: r_ms(a) + 8       */
    double rms;             /* inner disk edge used for the r >= rms test; <= 0: r_ms(a) */
    double bh_mass;         /* disk_nt_setup arguments (M [Msun], mdot [Edd], alpha)     */
    double mdot;
    double alpha_visc;
    int    max_order;       /* number of equatorial crossings tried (reference: 2)       */
    int    flags;           /* SIM5GPU_IMG_*                                             */
    double pol_degree;      /* polarization degree delta of the disk emission (polarized) */
    /* Optional striping for multi-GPU sharding: with stripe_rows > 0 the call traces the rows
     * [y0 + j*stripe_step, y0 + j*stripe_step + stripe_rows) for j = 0, 1, ... below y1, in ONE launch;
     * the outputs hold those rows packed in that order.  stripe_rows == 0: the contiguous range.     */
    int    stripe_rows;
    int    stripe_step;
    double disk_spin;       /* spin the disk model was set up with, if it differs from `a` (callers that
                               clamp the ray-tracing spin, ref python/sim5diskraytrace.py:32); < 0: use a */
} sim5gpu_image_desc;

#define SIM5GPU_IMG_DEFAULT 0   /* tuned FP64 sequences ("fast" variant, sim5_amd/csrc/s5_config.hpp)        */
#define SIM5GPU_IMG_STRICT  1   /* reference parameters, IEEE sqrt/div, no FMA contraction ("strict")       */
#define SIM5GPU_IMG_MIRROR  2   /* the call also traces the mirror image ny-1-y of every row y it traces.  The rows
                                 * named by y0, y1 (and the striping) must lie in the upper half, y1 <= (ny+1)/2; the
                                 * outputs hold ALL rows of the call in increasing row order (the named rows, then
                                 * their mirrors); the middle row of an odd ny is its own mirror and appears once.  A ray
                                 * and its mirror image in beta share the geodesic (same l, q: ref src/sim5kerr-geod.c:
                                 * 76-77), so the default variant traces such a pair in one lane at about two thirds of
                                 * the cost of two rays; a multi-GPU split deals mirrored stripe pairs for that reason
                                 * (sim5_amd/sharding.py).  A plain call whose range is symmetric (y0 + y1 == ny) is paired
                                 * the same way without asking.  The values are those of the unpaired trace, bit for bit. */

#define SIM5GPU_IMG_INPLACE 4   /* sim5gpu_disk_image only: the output pointers are WHOLE-image planes (ny x nx) and every traced row
                                 * is written at its image row instead of packed -- a rank that assembles the image (the root of
                                 * a gather) traces its own stripes and its band straight into the final image.  The aux planes,
                                 * if given, are whole-image planes too. */
#define SIM5GPU_IMG_DIRECT  8   /* fast variant only: every ray takes the reference's own sequence (radial integral Rpc by R_F, the
                                   comparisons of P with Rpc and 2 Rpc, ref src/sim5kerr-geod.c:303-352, :881) with the fast arithmetic,
                                   instead of the addition-theorem form of r(P).  The default routine hands a few rays per million to
                                   that sequence by itself (sim5_amd/csrc/s5_thindisk.hpp); the flag is how a caller -- or a test --
                                   runs a whole image through it.  Same classes, values within rounding; ~40 % slower (0.44 against 0.31 ms at 4096^2).              */

/* optional full-precision outputs (any pointer may be NULL) */
typedef struct sim5gpu_image_aux {
    uint8_t *cls;           /* SIM5GPU_PX_*                         */
    int8_t  *gtype;         /* geodesic type, -1 if rejected        */
    double  *r;             /* radius of the accepted crossing, NaN */
    double  *g;             /* g-factor (0 if no hit)               */
    double  *flux;          /* local disk flux F(r) (0 if no hit)   */
} sim5gpu_image_aux;

/* rows of output a job description produces (y1 - y0, or the total height of its stripes) */
int sim5gpu_image_rows(const sim5gpu_image_desc *desc);

/* image row of every packed output row of a job description: rows[i], i < min(sim5gpu_image_rows(desc), capacity).
 * Host arithmetic (no GPU needed): the rule by which a share's rows go back into the image. */
int sim5gpu_image_row_map(const sim5gpu_image_desc *desc, int *rows, int capacity);

/* Put the packed rows of up to 16 shares of ONE image (e.g. the per-rank payloads a gather delivered to the root)
 * at their image rows, in one launch.  descs[i] is the job description share i was traced with; d_shares holds the
 * shares as consecutive blocks of [2 planes][share_rows][nx] floats (share_rows >= rows of the largest share: the
 * layout of gathered fixed-size payloads); d_image_f / d_image_g are the ny x nx planes of the whole image. */
int sim5gpu_image_place_shares(int n_shares, const sim5gpu_image_desc *descs, const float *d_shares, size_t share_rows,
                               float *d_image_f, float *d_image_g, void *stream);

/* A job description checked as the image launchers check it (geometry, striping, mirror rows, disk parameters): host
 * arithmetic only, no GPU.  0 or SIM5GPU_E_ARG with the reason in sim5gpu_last_error(). */
int sim5gpu_image_desc_check(const sim5gpu_image_desc *desc);

/* What a job description resolves to on the host before any ray is traced: rmax (the default r_ms(a) + 8 of ref
 * examples/04-disk-image-eqplane/disk-image.c:41-42 when desc->rmax <= 0), rms, and sin / cos of the inclination (NULL: not
 * wanted).  Host arithmetic only, no GPU; the values are the reference binary's bit for bit -- its r_ms, and the sincos() its
 * geodesic_init_inf calls: an ulp of rmax is an ulp of every alpha and beta, an ulp of cos i reaches Carter's constant. */
int sim5gpu_image_view(const sim5gpu_image_desc *desc, double *rmax, double *rms, double *sin_i, double *cos_i);

/* PCI bus id of a HIP device as text ("0000:05:00.0", len >= 16): lets the ranks of a multi-process job show that
 * they run on distinct GPUs. */
int sim5gpu_device_bus_id(int device, char *buf, int len);

/* Peer-to-peer form of the multi-GPU exchange ("Alternative without RCCL: ... into a root buffer"): the root
 * process exports the inter-process handle of its whole-image allocation (a hipMalloc'ed block: d_ptr must be the start of
 * the allocation), the peers map it and launch their shares with SIM5GPU_IMG_INPLACE on the mapped planes -- their rows go
 * straight into the root's image over xGMI, no payload buffer, no gather, no placement pass.  `handle` is
 * SIM5GPU_IPC_HANDLE_BYTES opaque bytes to be carried to the peers by any means; a mapped pointer is valid until
 * sim5gpu_ipc_close.  Completion is the caller's to signal (stream synchronisation + a barrier of its own). */
#define SIM5GPU_IPC_HANDLE_BYTES 64
int sim5gpu_ipc_export(const void *d_ptr, void *handle);
int sim5gpu_ipc_open(const void *handle, void **d_ptr);
int sim5gpu_ipc_close(void *d_ptr);

/* Self-check utility of the multi-GPU assembly: the number of 32-bit words in which two DEVICE buffers differ (bit
 * comparison, synchronous on the default stream) -- an assembled image against a single-launch one without a 134 MB
 * copy to the host.  No counterpart in the reference (its images live in host memory: memcmp). */
int sim5gpu_words_differ(const void *d_a, const void *d_b, size_t n_words, unsigned long long *h_count);

/* The caller loop of ref examples/04-disk-image-eqplane/disk-image.c:53-105 as one kernel:
 * image_f = (float)(F g^4), image_g = (float)g, zero where the ray does not hit the disk. */
int sim5gpu_disk_image(const sim5gpu_image_desc *desc, float *d_image_f, float *d_image_g,
                       const sim5gpu_image_aux *d_aux, void *stream);

/* n_jobs image jobs (descs[j] into d_image_f[j], d_image_g[j]; no full-precision planes) with as few launches as the jobs
 * allow: jobs of the default variant whose rows are symmetric about the middle of the image (whole images, centred bands,
 * SIM5GPU_IMG_MIRROR shares) run through ONE job-list launch per 16 jobs -- the jobs stream through the GPU back to back, so
 * small images (the caller loop of ref disk-image.c:53-105 over several spins or inclinations, or a rank's share of a split
 * image) do not pay a launch ramp and a ragged last round each; other jobs are launched by themselves, in order.  Every image
 * is what sim5gpu_disk_image gives for its job, bit for bit.  Everything is validated before the first launch. */
int sim5gpu_disk_image_jobs(int n_jobs, const sim5gpu_image_desc *descs, float *const *d_image_f, float *const *d_image_g,
                            void *stream);

/* Same job on caller-owned HOST buffers (allocates, launches, copies back, frees). */
int sim5gpu_disk_image_host(const sim5gpu_image_desc *desc, float *h_image_f, float *h_image_g,
                            const sim5gpu_image_aux *h_aux);

/* Same per-ray work for an explicit list of n rays (alpha[], beta[] in DEVICE memory,
 * SoA, read coalesced) instead of the implicit pixel grid. */
int sim5gpu_disk_rays(const sim5gpu_image_desc *desc, size_t n, const double *d_alpha,
                      const double *d_beta, float *d_image_f, float *d_image_g,
                      const sim5gpu_image_aux *d_aux, void *stream);

/* Thin-disk image with Walker-Penrose polarization transport: Stokes I = F g^4,
 * Q = delta I cos 2chi, U = delta I sin 2chi with chi from polarization_angle_rotation of the
 * Walker-Penrose constant of the local polarization vector at the emitter (recipe built from
 * ref src/sim5kerr-geod.c:787, src/sim5kerr.c:766,926,948,553, src/sim5polarization.c:145,272).
 * d_stokes: 3 planes [I | Q | U], each (y1-y0) x nx doubles; d_chi optional. */
int sim5gpu_disk_image_polarized(const sim5gpu_image_desc *desc, double *d_stokes,
                                 double *d_chi, const sim5gpu_image_aux *d_aux, void *stream);

/* Intersection of rays from infinity with the photosphere of a geometrically thick disk: the surface
 * search of the reference's Python ray tracer (python/sim5diskraytrace.py:214-335, geodesic(flat=False) and
 * __find_surface), one lane per ray.  The surface is H(R) given as a table (R ascending, n_table <= 4096
 * points, DEVICE memory), interpolated linearly, H = H[0] below the first point and a constant opening angle
 * beyond the last.  Outputs (DEVICE, n each; k is n x 4 and may be NULL): position integral P, radius r,
 * cos(theta) m, photon momentum k (pointing away from the disk, as :250) and status 1 = found / 0 = none.
 * strict: bit 0 selects the reference-parameter arithmetic; SIM5GPU_SURFACE_TABLE_CHECKED: the caller vouches that d_R
 * is strictly ascending and NaN-free.  Without it the entry point reads the table back and checks it -- a bad table is
 * an argument error of the call -- which waits for everything enqueued on `stream` before; with it the job is purely
 * asynchronous (an unchecked bad table gives status 0 / NaN rays, never a fault: every table access is bounded). */
#define SIM5GPU_SURFACE_TABLE_CHECKED 2
int sim5gpu_disk_surface_rays(double a, double incl, int n_table, const double *d_R, const double *d_H,
                              size_t n, const double *d_alpha, const double *d_beta,
                              double *d_P, double *d_r, double *d_m, double *d_k, int *d_status,
                              int strict, void *stream);

/* The same search plus the local frame of the disk surface at the point found: g = E_inf / E_local, the cosine
 * of the emission angle and the local flux, i.e. what the reference's DiskRaytrace.image() derives per pixel
 * with __tetrad, __gfactor and __emission_angle (python/sim5diskraytrace.py:176-198, 340-390) for a disk model
 * with this surface (dhdr = slope of the table), Novikov-Thorne flux and angular momentum for
 * (bh_mass, mdot, disk_spin; disk_spin < 0 = the hole's spin) and radial velocity d_vr[] on the nodes of the
 * table (NULL = 0).  d_g, d_mue, d_flux are NaN where status = 0; g <= 0 is returned as 0 (ref :360). */
int sim5gpu_disk_surface_frame(double a, double incl, double bh_mass, double mdot, double disk_spin,
                               int n_table, const double *d_R, const double *d_H, const double *d_vr,
                               size_t n, const double *d_alpha, const double *d_beta,
                               double *d_P, double *d_r, double *d_m, double *d_k, int *d_status,
                               double *d_g, double *d_mue, double *d_flux, int strict, void *stream);

/* Observed spectrum of the thin disk over the pixel grid of `desc` (first-order crossings, as the
 * reference's Python ray tracer: python/sim5diskraytrace.py:96-123, black body of
 * python/sim5diskspectrum.py:54-88): spectrum[j] = sum over pixels of I_nu(E_j / g) g^3, with T_eff from the
 * Novikov-Thorne flux, g and the emission angle from the local frame of the disk surface.  The caller
 * multiplies by the solid angle of a pixel.  d_energies [keV] and d_spectrum hold n_energies doubles in
 * DEVICE memory; d_workspace needs sim5gpu_disk_spectrum_workspace(desc, n_energies) bytes. */
size_t sim5gpu_disk_spectrum_workspace(const sim5gpu_image_desc *desc, int n_energies);
int sim5gpu_disk_spectrum(const sim5gpu_image_desc *desc, int n_energies, const double *d_energies,
                          double hardening, int limb_darkening, double *d_spectrum,
                          void *d_workspace, void *stream);

/* Step-wise (Verlet) ray tracer with radiative transfer through an optically thin torus.
 * Rays start on the incoming branch at radius r0 (geodesic_init_inf -> geodesic_P_int ->
 * geodesic_position_pol -> geodesic_momentum), are advanced by raytrace() until they leave
 * [r_stop_in * r_bh, r_stop_out * r0], exceed max_steps or rtd.error > max_error, and
 * accumulate dI = g^4 j e^-tau dl, dtau = kappa_abs j-weighted dl in the torus. */
typedef struct sim5gpu_torus_desc {
    sim5gpu_image_desc img;       /* image geometry (disk fields unused)                */
    double r0;                    /* starting radius of the integration [GM/c^2]        */
    double dl_max;                /* cap handed to raytrace() in *step (<= 0: 1e9)       */
    double precision;             /* raytrace_prepare precision factor                   */
    int    options;               /* SIM5GPU_RTOPT_*                                      */
    int    max_steps;
    double max_error;             /* stop when rtd.error exceeds this (reference: 1e-2)  */
    double r_stop_in, r_stop_out; /* in units of r_bh and r0                              */
    int    shape;                 /* 0: Gaussian torus; 1: uniform sphere of radius torus_w */
    double torus_r, torus_w;      /* centre radius and Gaussian width of the torus        */
    double torus_l;               /* specific angular momentum of the torus fluid         */
    double emis0, absorb0;        /* emissivity and absorption normalisations             */
} sim5gpu_torus_desc;

typedef struct sim5gpu_torus_aux {
    int    *steps;           /* raytrace() calls made                        */
    float  *max_step_error;  /* largest rtd.error seen                       */
    double *carter_error;    /* raytrace_error() at the end                  */
    double *x_end;           /* n x 4 final position                         */
    double *k_end;           /* n x 4 final momentum                         */
} sim5gpu_torus_aux;

/* The surface-search and torus jobs keep a per-device workspace that grows to the largest job seen (surface search:
 * ~0.5 GB per million rays; torus: 120 B per ray), and the image jobs of the default variant keep one 8 KB block per disk
 * model (spin, mdot / M) and device -- the Novikov-Thorne flux table -- for up to 1024 models per device (least recently
 * used first out).  This gives all of it back (waits for the devices that own it); the next job allocates again.
 * *bytes, if not NULL, receives the number of bytes freed.  NOT to be called while another thread is inside, or about to
 * launch, a job of this library: a job that has been handed a table block must have been launched before the block goes
 * (the call waits for launched work, it cannot wait for a thread that has not launched yet). */
int sim5gpu_release_workspaces(size_t *bytes);

/* d_stokes: (y1-y0) x nx records of sim5gpu_stokes */
int sim5gpu_torus_image(const sim5gpu_torus_desc *desc, sim5gpu_stokes *d_stokes,
                        const sim5gpu_torus_aux *d_aux, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SIM5GPU_H */
