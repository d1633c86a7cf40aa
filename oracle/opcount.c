/*
 * opcount.c -- exact dynamic floating-point operation counts of the UNMODIFIED reference binary
 * (oracle/_ref/libsim5ref.so) on the hot path, per ray and per raytrace() step.
 *
 * TEST / MEASUREMENT INFRASTRUCTURE ONLY ("replace these estimates by exact counts").
 *
 * How: a child process runs the caller loop of the reference's example (ref:
 * examples/04-disk-image-eqplane/disk-image.c:53-105) or a raytrace() loop (ref: src/sim5raytrace.c:109)
 * on a sample of rays and brackets each ray with SIGUSR1 / SIGUSR2; the parent single-steps it with ptrace
 * between the two signals and classifies every instruction executed inside the reference library's text by
 * its opcode bytes (SSE2 scalar/packed double arithmetic, compares, min/max, x87 for the long-double branch
 * of geodesic_priv_T_roots, float ops of the disk-nt statics).  A call that leaves the library (libm through
 * the PLT) counts as ONE library call and is stepped over with a temporary breakpoint at its return
 * address.  Nothing is estimated: the numbers are what this build of the reference executes.
 *
 *   opcount <libsim5ref.so> image  <n> <a> <inc_deg>          n x n sample grid of the image plane
 *   opcount <libsim5ref.so> verlet <n> <a> <inc_deg> <prec>   n x n rays, r0 = 100, first 200 steps each
 *   opcount <libsim5ref.so> surface <n> <a> <inc_deg>         n x n rays of the thick-disk surface search (bench.py f1)
 *
 * Output: one JSON object on stdout.  OPCOUNT_TARGETS=1 additionally lists the address of every library call
 * on stderr (on this image: fmax 68 and fmin 19 per ray -- gcc does not inline them without -ffinite-math --
 * then csqrt, log, pow, sincos, acos, atan2, cos).
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <errno.h>
#include <fcntl.h>
#include <math.h>
#include <signal.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/ptrace.h>
#include <sys/user.h>
#include <sys/wait.h>
#include <unistd.h>

/* ---- layouts of the reference's structs (ref: src/sim5kerr-geod.h:42-68, src/sim5raytrace.h:26-43) ---- */
typedef struct { double v[30]; } geodesic_blob;          /* 240 B */
typedef struct { char b[144]; } rtd_blob;                /* 144 B; error (float) at offset 136 */

enum { C_ADD, C_SUB, C_MUL, C_DIV, C_SQRT, C_CMP, C_MINMAX, C_PACKED, C_LOGIC, C_CVT, C_F32, C_X87, C_LIBCALL, C_OTHER, C_N };
static const char *cname[C_N] = { "add", "sub", "mul", "div", "sqrt", "compare", "minmax", "packed_pd_arith", "sign_logic",
                                  "convert", "f32_arith", "x87", "library_calls", "other_instructions" };

static int classify(const unsigned char *p)
{
    int i = 0, pfx = 0;
    for (;; ++i) {                       /* legacy prefixes */
        if (p[i] == 0x66 || p[i] == 0xF2 || p[i] == 0xF3) { pfx = p[i]; continue; }
        if (p[i] == 0x2E || p[i] == 0x3E || p[i] == 0x26 || p[i] == 0x36 || p[i] == 0x64 || p[i] == 0x65 || p[i] == 0x67) continue;
        break;
    }
    if ((p[i] & 0xF0) == 0x40) ++i;      /* REX */
    if (p[i] >= 0xD8 && p[i] <= 0xDF) return C_X87;
    if (p[i] != 0x0F) return C_OTHER;
    const unsigned char op = p[i + 1];
    if (pfx == 0xF2) {
        switch (op) {
        case 0x58: return C_ADD; case 0x5C: return C_SUB; case 0x59: return C_MUL; case 0x5E: return C_DIV;
        case 0x51: return C_SQRT; case 0x5D: case 0x5F: return C_MINMAX; case 0xC2: return C_CMP;
        case 0x2A: case 0x2C: case 0x2D: case 0x5A: return C_CVT;
        default: return C_OTHER;
        }
    }
    if (pfx == 0x66) {
        switch (op) {
        case 0x58: case 0x5C: case 0x59: case 0x5E: case 0x51: case 0x5D: case 0x5F: return C_PACKED;
        case 0x2E: case 0x2F: return C_CMP;
        case 0x54: case 0x55: case 0x56: case 0x57: return C_LOGIC;
        case 0x5A: return C_CVT;
        default: return C_OTHER;
        }
    }
    if (pfx == 0xF3) {
        switch (op) {
        case 0x58: case 0x5C: case 0x59: case 0x5E: case 0x51: case 0x5D: case 0x5F: return C_F32;
        case 0x5A: case 0x2A: case 0x2C: case 0x2D: case 0xE6: return C_CVT;
        default: return C_OTHER;
        }
    }
    if (pfx == 0) {
        switch (op) {
        case 0x2E: case 0x2F: return C_F32;                 /* ucomiss / comiss */
        case 0x54: case 0x55: case 0x56: case 0x57: return C_LOGIC;
        case 0x5A: return C_CVT;
        default: return C_OTHER;
        }
    }
    return C_OTHER;
}

/* ---- child: the measured workloads ---- */
typedef int (*fn_init_inf)(double, double, double, double, void *, int *);
typedef double (*fn_gd_i)(void *, int);
typedef double (*fn_gd_d)(void *, double);
typedef double (*fn_ddd)(double, double, double);
typedef double (*fn_d)(double);
typedef int (*fn_setup)(double, double, double, double, int);
typedef double (*fn_Pint)(void *, double, int);
typedef void (*fn_mom)(void *, double, double, double, double *);
typedef void (*fn_prep)(double, double *, double *, double, int, void *);
typedef void (*fn_rt)(double *, double *, double *, void *);

static volatile double sink;

static void child_image(void *lib, int n, double a, double inc)
{
    fn_init_inf init_inf = (fn_init_inf)dlsym(lib, "geodesic_init_inf");
    fn_gd_i crossing = (fn_gd_i)dlsym(lib, "geodesic_find_midplane_crossing");
    fn_gd_d position_rad = (fn_gd_d)dlsym(lib, "geodesic_position_rad");
    fn_ddd gfactorK = (fn_ddd)dlsym(lib, "gfactorK");
    fn_d flux = (fn_d)dlsym(lib, "disk_nt_flux"), r_ms = (fn_d)dlsym(lib, "r_ms");
    fn_setup setup = (fn_setup)dlsym(lib, "disk_nt_setup");
    setup(10.0, a, 0.1, 0.1, 0);
    const double rms = r_ms(a), rmax = rms + 8.0;
    for (int iy = 0; iy < n; iy++) for (int ix = 0; ix < n; ix++) {
        const double alpha = (((double)ix + .5) / (double)n - 0.5) * 2.0 * rmax;
        const double beta = (((double)iy + .5) / (double)n - 0.5) * 2.0 * rmax;
        raise(SIGUSR1);
        /* body of the pixel loop, ref: disk-image.c:57-100 */
        geodesic_blob gd; int err;
        if (init_inf(inc, a, alpha, beta, &gd, &err)) {
            double P = crossing(&gd, 0);
            if (!isnan(P)) {
                double r = position_rad(&gd, P);
                if (r < rms) {
                    P = crossing(&gd, 1);
                    r = isnan(P) ? 0.0 : position_rad(&gd, P);
                }
                if (r >= rms) {
                    const double g = gfactorK(r, a, gd.v[5]);
                    const double f = flux(r);
                    sink = f * pow(g, 4.);
                }
            }
        }
        raise(SIGUSR2);
    }
}

static void child_verlet(void *lib, int n, double a, double inc, double precision)
{
    fn_init_inf init_inf = (fn_init_inf)dlsym(lib, "geodesic_init_inf");
    fn_Pint P_int = (fn_Pint)dlsym(lib, "geodesic_P_int");
    fn_gd_d position_pol = (fn_gd_d)dlsym(lib, "geodesic_position_pol");
    fn_mom momentum = (fn_mom)dlsym(lib, "geodesic_momentum");
    fn_prep prepare = (fn_prep)dlsym(lib, "raytrace_prepare");
    fn_rt raytrace = (fn_rt)dlsym(lib, "raytrace");
    fn_d r_ms = (fn_d)dlsym(lib, "r_ms"), r_bh = (fn_d)dlsym(lib, "r_bh");
    const double rmax = r_ms(a) + 8.0, r0 = 100.0, rh = r_bh(a);
    for (int iy = 0; iy < n; iy++) for (int ix = 0; ix < n; ix++) {
        const double alpha = (((double)ix + .5) / (double)n - 0.5) * 2.0 * rmax;
        const double beta = (((double)iy + .5) / (double)n - 0.5) * 2.0 * rmax;
        geodesic_blob gd; int err;
        if (!init_inf(inc, a, alpha, beta, &gd, &err)) continue;
        if (!(r0 > gd.v[20])) continue;                          /* rp */
        const double P0 = P_int(&gd, r0, 0);
        double x[4] = { 0.0, r0, position_pol(&gd, P0), 0.0 }, k[4];
        momentum(&gd, P0, x[1], x[2], k);
        if (isnan(k[0]) || isnan(x[2])) continue;
        rtd_blob rtd;
        prepare(a, x, k, precision, 0, &rtd);
        /* the steps of a ray are sampled: every 7th call is measured */
        for (int step = 0; step < 4000; step++) {
            double dl = 1e9;
            const int measured = (step % 7) == 3;
            if (measured) raise(SIGUSR1);
            raytrace(x, k, &dl, &rtd);
            if (measured) raise(SIGUSR2);
            float e; memcpy(&e, rtd.b + 136, 4);
            if (x[1] < 1.05 * rh || x[1] > 1.01 * r0 || e > 1e-2f) break;
        }
    }
}

/* The surface search of the reference's Python ray tracer (ref python/sim5diskraytrace.py:257-335: DiskRaytrace.__find_surface),
 * statement for statement, over the reference's C functions; what is counted is what those functions execute (the Python
 * arithmetic between the calls -- the step rule, H(R) of the disk model -- is not library text, here as there).  The job of
 * bench.py's f1 line: thick disk H(R) = 0.25 (R - 2) on the table range 2 <= R <= 60 (held constant outside it), a, i,
 * n x n rays over a field of view of +-20. */
typedef void (*fn_follow)(void *, double, double *, double *, double *, int *);
static double disk_h(double R) { const double Rc = R < 2.0 ? 2.0 : (R > 60.0 ? 60.0 : R); return 0.25 * (Rc - 2.0); }

static int find_surface(void *gd, int iteration, double a, fn_Pint P_int, fn_gd_d position_rad, fn_gd_d position_pol,
                        fn_follow follow, fn_gd_i crossing, double rbh)
{
    if (iteration > 3) return 0;
    const double *g = (const double *)gd;                      /* a@0 alpha@8 beta@16 incl@24 ... rp@160 */
    const double alpha = g[1], beta = g[2], incl = g[3], rp = g[20];
    const double disk_theta = atan(disk_h(1e6) / 1e6);
    double r0 = fmax(200.0, fmax(1.1 * rp, (0.5 + iteration) * sqrt(alpha * alpha + beta * beta) / cos(incl + disk_theta)));
    const double accuracy = 1e-2;
    double P1, r1, m1, H1, Hd;
    for (;;) {
        P1 = P_int(gd, r0, 0); r1 = position_rad(gd, P1); m1 = position_pol(gd, P1);
        H1 = r1 * m1; Hd = disk_h(r1 * sqrt(1. - m1 * m1));
        if (Hd < H1 || r0 > 5e6) break;
        r0 = 2.0 * r0;
    }
    if (Hd >= H1) return 0;
    double P = P1, r = r1, m = m1, step_factor = 1.0;
    int status = 0;
    for (;;) {
        const double step = fmax(accuracy / 2., fmin((H1 - Hd) / 2., 0.5 * (sqrt(r) - 0.99) * step_factor));
        follow(gd, step, &P, &r, &m, &status);
        if (!status) return 0;
        H1 = r * m; Hd = disk_h(r * sqrt(1. - m * m));
        if (H1 <= Hd) {
            if (step < accuracy) { follow(gd, -step / 2., &P, &r, &m, &status); return 1; }
            follow(gd, -step, &P, &r, &m, &status);
            step_factor = step_factor / 5.;
            continue;
        }
        if (H1 < 1e-4) { const double Pc = crossing(gd, 0); sink = position_rad(gd, Pc) + position_pol(gd, Pc); return 1; }
        if (r < 1.05 * rbh) return 0;
        if (r > 1.1 * r0) return find_surface(gd, iteration + 1, a, P_int, position_rad, position_pol, follow, crossing, rbh);
        if (m < 0.0) return 0;
        if (step < accuracy / 2.) break;
    }
    return 0;
}

static void child_surface(void *lib, int n, double a, double inc)
{
    fn_init_inf init_inf = (fn_init_inf)dlsym(lib, "geodesic_init_inf");
    fn_Pint P_int = (fn_Pint)dlsym(lib, "geodesic_P_int");
    fn_gd_d position_rad = (fn_gd_d)dlsym(lib, "geodesic_position_rad"), position_pol = (fn_gd_d)dlsym(lib, "geodesic_position_pol");
    fn_follow follow = (fn_follow)dlsym(lib, "geodesic_follow");
    fn_gd_i crossing = (fn_gd_i)dlsym(lib, "geodesic_find_midplane_crossing");
    fn_d r_bh = (fn_d)dlsym(lib, "r_bh");
    const double rmax = 20.0, rbh = r_bh(a);
    long hits = 0;
    for (int iy = 0; iy < n; iy++) for (int ix = 0; ix < n; ix++) {
        const double alpha = (((double)ix + .5) / (double)n - 0.5) * 2.0 * rmax;
        const double beta = (((double)iy + .5) / (double)n - 0.5) * 2.0 * rmax;
        raise(SIGUSR1);
        geodesic_blob gd; int err;
        if (init_inf(inc, a, alpha, beta, &gd, &err) && !err)                     /* ref py :228-243 */
            hits += find_surface(&gd, 0, a, P_int, position_rad, position_pol, follow, crossing, rbh);
        raise(SIGUSR2);
    }
    sink = (double)hits;
}

/* ---- parent ---- */
static int text_range(pid_t pid, const char *needle, uintptr_t *lo, uintptr_t *hi)
{
    char path[64]; snprintf(path, sizeof path, "/proc/%d/maps", (int)pid);
    FILE *f = fopen(path, "r"); if (!f) return -1;
    char line[1024]; int found = 0;
    while (fgets(line, sizeof line, f)) {
        unsigned long a, b; char perm[8];
        if (sscanf(line, "%lx-%lx %7s", &a, &b, perm) != 3) continue;
        if (strstr(line, needle) && perm[2] == 'x') { *lo = a; *hi = b; found = 1; }
    }
    fclose(f);
    return found ? 0 : -1;
}

#define CACHE_BITS 16
static struct { uintptr_t rip; int cls; } cache[1 << CACHE_BITS];

static int class_at(pid_t pid, uintptr_t rip)
{
    const unsigned h = (unsigned)((rip * 0x9E3779B97F4A7C15ull) >> (64 - CACHE_BITS));
    if (cache[h].rip == rip) return cache[h].cls;
    unsigned char buf[16];
    for (int i = 0; i < 2; i++) {
        errno = 0;
        long w = ptrace(PTRACE_PEEKTEXT, pid, (void *)(rip + 8 * i), 0);
        if (errno) w = 0;
        memcpy(buf + 8 * i, &w, 8);
    }
    cache[h].rip = rip; cache[h].cls = classify(buf);
    return cache[h].cls;
}

int main(int argc, char **argv)
{
    if (argc < 6) { fprintf(stderr, "usage: opcount <lib> image|verlet|surface <n> <a> <inc_deg> [precision]\n"); return 2; }
    const char *libpath = argv[1]; const int verlet = !strcmp(argv[2], "verlet"), surface = !strcmp(argv[2], "surface");
    const int n = atoi(argv[3]); const double a = atof(argv[4]), inc = atof(argv[5]) / 180.0 * M_PI;
    const double precision = argc > 6 ? atof(argv[6]) : 1.0;

    pid_t pid = fork();
    if (pid == 0) {
        void *lib = dlopen(libpath, RTLD_NOW | RTLD_LOCAL);
        if (!lib) { fprintf(stderr, "opcount: %s\n", dlerror()); _exit(3); }
        int devnull = open("/dev/null", 1); dup2(devnull, 2);          /* the reference's diagnostics */
        ptrace(PTRACE_TRACEME, 0, 0, 0);
        raise(SIGSTOP);
        if (verlet) child_verlet(lib, n, a, inc, precision); else if (surface) child_surface(lib, n, a, inc); else child_image(lib, n, a, inc);
        _exit(0);
    }
    int st; waitpid(pid, &st, 0);
    uintptr_t lo = 0, hi = 0;
    const char *base = strrchr(libpath, '/'); base = base ? base + 1 : libpath;
    if (getenv("OPCOUNT_TARGETS")) { char cmd[128]; snprintf(cmd, sizeof cmd, "cat /proc/%d/maps | grep r-xp >&2", (int)pid); (void)!system(cmd); }
    if (text_range(pid, base, &lo, &hi)) { fprintf(stderr, "opcount: cannot find %s in the child's maps\n", base); kill(pid, SIGKILL); return 3; }

    unsigned long long tot[C_N] = { 0 }, units = 0, instr_in_lib = 0;
    int measuring = 0, prev_in_lib = 0, prev_was_ret = 0;
    ptrace(PTRACE_CONT, pid, 0, 0);
    for (;;) {
        waitpid(pid, &st, 0);
        if (WIFEXITED(st) || WIFSIGNALED(st)) { if (WIFSIGNALED(st) || WEXITSTATUS(st)) fprintf(stderr, "opcount: child ended abnormally (status 0x%x)\n", st); break; }
        const int sig = WSTOPSIG(st);
        if (sig == SIGUSR1) { measuring = 1; prev_in_lib = 0; units++; ptrace(PTRACE_SINGLESTEP, pid, 0, 0); continue; }
        if (sig == SIGUSR2) { measuring = 0; ptrace(PTRACE_CONT, pid, 0, 0); continue; }
        if (sig != SIGTRAP) { ptrace(measuring ? PTRACE_SINGLESTEP : PTRACE_CONT, pid, 0, sig); continue; }
        if (!measuring) { ptrace(PTRACE_CONT, pid, 0, 0); continue; }
        struct user_regs_struct regs;
        ptrace(PTRACE_GETREGS, pid, 0, &regs);
        const uintptr_t rip = regs.rip;
        if (rip >= lo && rip < hi) {
            tot[class_at(pid, rip)]++;
            instr_in_lib++;
            prev_in_lib = 1;
            prev_was_ret = 0;
            {   /* remember whether this instruction is a return (C3, or F3 C3) */
                errno = 0;
                const long w = ptrace(PTRACE_PEEKTEXT, pid, (void *)rip, 0);
                const unsigned char b0 = (unsigned char)(w & 0xFF), b1 = (unsigned char)((w >> 8) & 0xFF);
                if (!errno && (b0 == 0xC3 || (b0 == 0xF3 && b1 == 0xC3))) prev_was_ret = 1;
            }
            ptrace(PTRACE_SINGLESTEP, pid, 0, 0);
            continue;
        }
        /* outside the reference library.  Arriving here from an instruction inside it that was not a return
           is a call out (libm through the PLT): ONE library call; its return address is on top of the stack,
           so run to it with a temporary breakpoint.  Everything else outside (the driver, raise()) is stepped. */
        if (prev_in_lib && !prev_was_ret) {
            tot[C_LIBCALL]++;
            if (getenv("OPCOUNT_TARGETS")) fprintf(stderr, "T %lx\n", (unsigned long)rip);
            errno = 0;
            const uintptr_t ret = (uintptr_t)ptrace(PTRACE_PEEKDATA, pid, (void *)regs.rsp, 0);
            if (!errno && ret >= lo && ret < hi) {
                const long orig = ptrace(PTRACE_PEEKTEXT, pid, (void *)ret, 0);
                const long patched = (orig & ~0xFFl) | 0xCC;
                ptrace(PTRACE_POKETEXT, pid, (void *)ret, (void *)patched);
                ptrace(PTRACE_CONT, pid, 0, 0);
                waitpid(pid, &st, 0);
                ptrace(PTRACE_POKETEXT, pid, (void *)ret, (void *)orig);
                if (WIFEXITED(st) || WIFSIGNALED(st)) break;
                if (WSTOPSIG(st) == SIGTRAP) {
                    ptrace(PTRACE_GETREGS, pid, 0, &regs);
                    if ((uintptr_t)regs.rip == ret + 1) { regs.rip = ret; ptrace(PTRACE_SETREGS, pid, 0, &regs); }
                }
            }
        }
        prev_in_lib = 0;
        ptrace(PTRACE_SINGLESTEP, pid, 0, 0);
    }
    const double u = units ? (double)units : 1.0;
    const double flops = (double)(tot[C_ADD] + tot[C_SUB] + tot[C_MUL] + tot[C_DIV] + tot[C_SQRT] + tot[C_CMP] + tot[C_MINMAX]
                                  + 2 * tot[C_PACKED] + tot[C_X87] + tot[C_LIBCALL]) / u;
    printf("{\"workload\": \"%s\", \"units\": %llu, \"unit\": \"%s\", \"spin\": %g, \"incl_deg\": %g, \"sample_grid\": %d",
           verlet ? "raytrace() calls, r0 = 100" : surface ? "surface search of a thick disk H(R) = 0.25 (R - 2), field of view +-20 (python/sim5diskraytrace.py:228-335 over the C library)" : "thin-disk pixel loop body (disk-image.c:57-100)", units,
           verlet ? "raytrace() call" : "ray", a, atof(argv[5]), n);
    if (verlet) printf(", \"precision\": %g", precision);
    printf(", \"per_unit\": {");
    for (int c = 0; c < C_N; c++) printf("%s\"%s\": %.2f", c ? ", " : "", cname[c], (double)tot[c] / u);
    printf("}, \"instructions_in_reference_text_per_unit\": %.1f", (double)instr_in_lib / u);
    printf(", \"fp64_ops_per_unit\": %.1f, \"convention\": \"add+sub+mul+div+sqrt+compare+minmax + 2*packed + x87 + 1 per library call (SURVEY.md 8(d))\"}\n", flops);
    return 0;
}
