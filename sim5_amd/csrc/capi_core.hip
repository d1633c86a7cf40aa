// capi_core.hip -- C-ABI: runtime helpers, disk model set-up and the whole-job image entry points
// declared in include/sim5gpu.h.  Host code here only validates arguments, folds per-image
// constants and launches kernels; there is no CPU implementation of any ray computation.
#include "capi_util.hpp"
#include <math.h>
#include <string.h>
#include <algorithm>
#include <atomic>
#include <mutex>
#include <unordered_map>
#include <vector>

namespace s5 {

thread_local char g_err[512] = "";
DiskConsts g_disk = {};            // process-global like the reference's statics (src/sim5disk-nt.c:27-32)

Arena& arena()
{
    static thread_local Arena a;
    return a;
}

namespace {
struct ThreadStream {
    hipStream_t s = nullptr;
    int dev = -1;
    ~ThreadStream() { if (s) (void)hipStreamDestroy(s); }
};
}

hipStream_t thread_stream()
{
    static thread_local ThreadStream t;
    int cur = 0;
    if (hipGetDevice(&cur) != hipSuccess) return nullptr;
    if (t.s && t.dev == cur) return t.s;
    if (t.s) { (void)hipStreamSynchronize(t.s); (void)hipStreamDestroy(t.s); t.s = nullptr; }
    if (hipStreamCreateWithFlags(&t.s, hipStreamNonBlocking) != hipSuccess) { t.s = nullptr; t.dev = -1; return nullptr; }   // (falls back to the null stream)
    t.dev = cur;
    return t.s;
}

void set_error(const char* what, hipError_t e)
{
    snprintf(g_err, sizeof g_err, "%s: %s", what, e == hipSuccess ? "" : hipGetErrorString(e));
}

int have_device()
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        snprintf(g_err, sizeof g_err, "no HIP device available (%s); libsim5gpu has no CPU fallback",
                 e == hipSuccess ? "device count is 0" : hipGetErrorString(e));
        return 0;
    }
    return n;
}

// ISCO radius on the host, the reference's r_ms (src/sim5kerr.c:994-1004)
static double host_r_ms(double a)
{
    // (3 (a a), as the reference's 3.*sqr(a) associates -- not (3 a) a, which is what disk_nt_r_min has: an ulp of r_ms is an
    // ulp of the default field of view, i.e. of every alpha and beta of the image)
    double z1 = 1. + cbrt(1. - a * a) * (cbrt(1. + a) + cbrt(1. - a));
    double z2 = sqrt(3. * (a * a) + z1 * z1);
    return 3. + z2 - sqrt((3. - z1) * (3. + z1 + 2. * z2));
}

// Fold everything that depends only on (M, a, mdot) -- float-rounded as the reference's statics are
// -- into DiskConsts (ref src/sim5disk-nt.c:37-78 setup, :91-105 r_min, :122-135 flux constants).
void disk_set_mdot(DiskConsts& d, double mdot)
{
    d.mdot = (float)mdot;                                  // the reference keeps it in a float static
    d.scale = 9.1721376255e+28 * d.mdot / d.mass;
}

DiskConsts make_disk_consts(double M, double a_in, double mdot, double alpha)
{
    DiskConsts d;
    const float f_mass = (float)M, f_spin = (float)a_in, f_mdot = (float)mdot, f_alpha = (float)alpha;
    const double a = f_spin;
    const double sga = (a >= 0.0) ? +1. : -1.;
    const double z1 = 1. + pow(1. - a * a, 1. / 3.) * (pow(1. + a, 1. / 3.) + pow(1. - a, 1. / 3.));
    const double z2 = sqrt(3. * a * a + z1 * z1);
    const double r0 = 3. + z2 - sga * sqrt((3. - z1) * (3. + z1 + 2. * z2));
    const float f_rms = (float)(r0 + 1e-3);
    d.a = a;
    d.rms = f_rms;
    d.x0 = sqrt((double)f_rms);
    d.x1 = +2. * cos(1. / 3. * acos(a) - M_PI / 3.);
    d.x2 = +2. * cos(1. / 3. * acos(a) + M_PI / 3.);
    d.x3 = -2. * cos(1. / 3. * acos(a));
    d.p1 = 3. * ((d.x1 - a) * (d.x1 - a)) / (d.x1 * (d.x1 - d.x2) * (d.x1 - d.x3));
    d.p2 = 3. * ((d.x2 - a) * (d.x2 - a)) / (d.x2 * (d.x2 - d.x1) * (d.x2 - d.x3));
    d.p3 = 3. * ((d.x3 - a) * (d.x3 - a)) / (d.x3 * (d.x3 - d.x1) * (d.x3 - d.x2));
    d.d1 = d.x0 - d.x1;
    d.d2 = d.x0 - d.x2;
    d.d3 = d.x0 - d.x3;
    d.mdot = f_mdot;
    d.mass = f_mass;
    d.inv_x0 = 1.0 / d.x0; d.inv_d1 = 1.0 / d.d1; d.inv_d2 = 1.0 / d.d2; d.inv_d3 = 1.0 / d.d3;
    d.scale = 9.1721376255e+28 * d.mdot / d.mass;
    d.alpha = f_alpha;
    d.a2f = (double)(f_spin * f_spin);
    d.ftab = nullptr; d.cold = nullptr; d.ft_wmin = 0.0; d.ft_inv_dw = 0.0;
    d.ready = 1;
    return d;
}

// ---- radial profile table of the Novikov-Thorne flux (fast variant) ---------------------------------------
// F(r) is a closed form with four logarithms (ref src/sim5disk-nt.c:110-146) and, per pixel, the second most expensive
// block of the image kernel after r(P).  Up to the scale mdot/M it is a function of the radius and of the spin only, so
// the host tabulates  T(w) = F / (scale (x - x0)),  x = sqrt(r), w = x0 / x  -- smooth and bounded on (0, 1]: F vanishes
// linearly at x0 = sqrt of the reference's float-rounded inner edge, which is 1e-3 outside the true ISCO -- as FT_N
// polynomials of degree FT_DEG in the local coordinate of FT_N equal intervals of w in [x0 / 16, 1]: Chebyshev
// interpolation of the reference's formula evaluated in long double, 1e-10 of T at a = 0.998, 3e-13 at a <= 0.9
// (checked in tests/test_gpu_images.py against the closed form on the device).  The kernels keep the closed form where
// the table must not be used: within 2e-4 of x0, where the reference's own double evaluation is a rounding-noise
// pattern that parity has to reproduce, and beyond x = 16.  8 KB per spin, kept per device for the last few spins.
namespace {
// One device block per (device, disk model): the COLD_N constants of the closed form (read from memory by the image
// kernels' rare closed-form lanes, so that they do not occupy ~30 SGPRs of every wave for the whole kernel), then the
// table; 8 KB each.  Found through a hash of (spin, scale) -- a fitting sweep over 1e5 disk models costs a look-up per
// launch, not a scan -- and BOUNDED: beyond FT_CACHE_MAX models per device the least recently used half is retired.
// A retired block is not freed at once (a thread that has just been handed its pointer may be about to launch with it):
// it rests until the NEXT retirement, FT_CACHE_MAX / 2 new models later, and is freed then, after the device has been
// waited for.  sim5gpu_release_workspaces frees everything (it waits for the device first).
constexpr int FT_DEVICES = 64;
constexpr size_t FT_CACHE_MAX = 1024;
struct FluxTable { double a, scale; double* ptr; bool usable; unsigned long long stamp; };
struct FluxKey { unsigned long long a, s; bool operator==(const FluxKey& o) const { return a == o.a && s == o.s; } };
struct FluxKeyHash { size_t operator()(const FluxKey& k) const { return (size_t)(k.a * 0x9e3779b97f4a7c15ull ^ (k.s + (k.a << 7))); } };
struct FluxCache {
    std::unordered_map<FluxKey, FluxTable, FluxKeyHash> live;
    std::vector<double*> resting;                   // retired at the last round: freed at the next, unless pinned
    std::unordered_map<const double*, int> pinned;  // blocks held by entry points between attach and the end of their launches
    unsigned long long clock = 0;
};
FluxCache g_ftab[FT_DEVICES];
std::mutex g_ftab_lock;
typedef std::vector<std::pair<int, const double*>> PinList;
thread_local FluxPinScope* t_pin_scope = nullptr;

static FluxKey flux_key(double a, double scale)
{
    a += 0.0; scale += 0.0;                          // (-0.0 and 0.0 are one model)
    FluxKey k;
    memcpy(&k.a, &a, sizeof a); memcpy(&k.s, &scale, sizeof scale);
    return k;
}

// called under the lock, on the device that owns the cache: the resting blocks nobody holds are handed to the caller, who
// frees them AFTER it has left the lock (a device synchronisation and hipFree under the lock stalled every other thread's launch)
static void flux_cache_retire(FluxCache& Cc, std::vector<double*>& to_free)
{
    if (Cc.live.size() < FT_CACHE_MAX) return;
    std::vector<double*> keep;
    for (double* p : Cc.resting) (Cc.pinned.count(p) ? keep : to_free).push_back(p);
    Cc.resting.swap(keep);
    std::vector<unsigned long long> stamps;
    stamps.reserve(Cc.live.size());
    for (auto& kv : Cc.live) stamps.push_back(kv.second.stamp);
    std::nth_element(stamps.begin(), stamps.begin() + stamps.size() / 2, stamps.end());
    const unsigned long long cut = stamps[stamps.size() / 2];
    for (auto it = Cc.live.begin(); it != Cc.live.end();) {
        if (it->second.stamp < cut) { Cc.resting.push_back(it->second.ptr); it = Cc.live.erase(it); }
        else ++it;
    }
}

// f on [lo, hi] as N polynomials of degree DEG in the local coordinate tau in [-1, 1] of N equal intervals: Chebyshev
// interpolation (DEG + 1 interior nodes per interval, long double), converted to monomials for a Horner evaluation on
// the device; tab[i * (DEG + 1) + q] = coefficient of tau^q in interval i.  Returns the largest relative error of the
// double-precision Horner evaluation, probed at four points inside every interval.
template <int N, int DEG, class Fn>
double fit_piecewise(Fn f, long double lo_all, long double hi_all, std::vector<double>& tab)
{
    typedef long double ld;
    const ld pi = 3.14159265358979323846264338327950288L;
    constexpr int K = DEG + 1;
    double worst = 0.0;
    tab.assign((size_t)N * K, 0.0);
    const ld dw = (hi_all - lo_all) / N;
    for (int i = 0; i < N; i++) {
        const ld lo = lo_all + dw * i, mid = lo + 0.5L * dw, half = 0.5L * dw;
        ld y[K], c[K];
        for (int k = 0; k < K; k++) y[k] = f(mid + half * cosl(pi * (k + 0.5L) / K));     // Chebyshev nodes (interior)
        for (int j = 0; j < K; j++) {                                                      // Chebyshev coefficients
            ld sacc = 0.L;
            for (int k = 0; k < K; k++) sacc += y[k] * cosl(pi * j * (k + 0.5L) / K);
            c[j] = sacc * (j == 0 ? 1.L : 2.L) / K;
        }
        // to monomials in tau: T_0 = 1, T_1 = tau, T_{n+1} = 2 tau T_n - T_{n-1}
        ld m[K] = { 0 }, tm1[K] = { 0 }, tm0[K] = { 0 }, tn[K];
        tm1[0] = 1.L; tm0[1] = 1.L;
        for (int q = 0; q < K; q++) m[q] += c[0] * tm1[q];
        if (K > 1) for (int q = 0; q < K; q++) m[q] += c[1] * tm0[q];
        for (int n = 2; n < K; n++) {
            for (int q = 0; q < K; q++) tn[q] = (q > 0 ? 2.L * tm0[q - 1] : 0.L) - tm1[q];
            for (int q = 0; q < K; q++) { m[q] += c[n] * tn[q]; tm1[q] = tm0[q]; tm0[q] = tn[q]; }
        }
        for (int q = 0; q < K; q++) tab[(size_t)i * K + q] = (double)m[q];
        for (int j = 0; j < 4; j++) {
            const ld tau = -0.9L + 0.6L * j;
            const ld w = mid + half * tau;
            if (w >= hi_all) continue;
            double acc = tab[(size_t)i * K + K - 1];
            for (int q = K - 2; q >= 0; q--) acc = acc * (double)tau + tab[(size_t)i * K + q];
            const ld ref = f(w);
            const double err = (double)fabsl(((ld)acc - ref) / ref);
            if (!(err <= worst)) worst = err;
        }
    }
    return worst;
}

// returns the largest relative error of the fit
double build_flux_table(const DiskConsts& d, double wmin, std::vector<double>& tab)
{
    typedef long double ld;
    const ld a = d.a, x0 = d.x0, x1 = d.x1, x2 = d.x2, x3 = d.x3;
    const ld p1 = 3.L * (x1 - a) * (x1 - a) / (x1 * (x1 - x2) * (x1 - x3));
    const ld p2 = 3.L * (x2 - a) * (x2 - a) / (x2 * (x2 - x1) * (x2 - x3));
    const ld p3 = 3.L * (x3 - a) * (x3 - a) / (x3 * (x3 - x1) * (x3 - x2));
    const ld pi = 3.14159265358979323846264338327950288L;
    auto T = [&](ld w) -> ld {
        const ld x = x0 / w;
        const ld f0 = x - x0 - 1.5L * a * logl(x / x0);
        const ld f1 = p1 * logl((x - x1) / (x0 - x1));
        const ld f2 = p2 * logl((x - x2) / (x0 - x2));
        const ld f3 = p3 * logl((x - x3) / (x0 - x3));
        const ld F = 1.L / (4.L * pi * x * x) * 1.5L / (x * x * (x * x * x - 3.L * x + 2.L * a)) * (f0 - f1 - f2 - f3);
        return F / (x - x0);
    };
    return fit_piecewise<s5abi::FT_N, s5abi::FT_DEG>(T, (long double)wmin, 1.L, tab);
}

// K(m), complete elliptic integral of the first kind, as KT_N polynomials of degree KT_DEG on [0, KT_MMAX]
// (the image kernels' fast variant; the AGM evaluation stays for larger moduli).  Universal: one table per device.
double build_K_table(std::vector<double>& tab)
{
    typedef long double ld;
    const ld pi = 3.14159265358979323846264338327950288L;
    auto K = [&](ld m) -> ld {
        ld a = 1.L, b = sqrtl(1.L - m);
        for (int it = 0; it < 40 && fabsl(a - b) > 1e-19L * a; it++) { const ld an = 0.5L * (a + b); b = sqrtl(a * b); a = an; }
        return pi / (a + b);
    };
    return fit_piecewise<s5abi::KT_N, s5abi::KT_DEG>(K, 0.L, (ld)s5abi::KT_MMAX, tab);
}
} // namespace

// universal K(m) table, one per device (never freed: 8 KB)
static const double* g_ktab[64];
static const double* g_sctab[64];

int attach_K_table(ImageParams& p)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return SIM5GPU_E_HIP;
    if (dev < 0 || dev >= 64) return SIM5GPU_OK;                       // no table: the kernels use the AGM
    std::lock_guard<std::mutex> hold(g_ftab_lock);
    if (!g_ktab[dev]) {
        std::vector<double> tab;
        const double fit_error = build_K_table(tab);
        if (!(fit_error <= 1e-15)) return SIM5GPU_OK;
        double* ptr = nullptr;
        hipError_t e = hipMalloc((void**)&ptr, tab.size() * sizeof(double));
        if (e == hipSuccess) e = hipMemcpy(ptr, tab.data(), tab.size() * sizeof(double), hipMemcpyHostToDevice);
        if (e != hipSuccess) { if (ptr) (void)hipFree(ptr); set_error("K table", e); return SIM5GPU_E_HIP; }
        g_ktab[dev] = ptr;
    }
    p.ktab = g_ktab[dev];
    // sin / cos nodes of msincos_tab (s5_trig.hpp), one block per device like the K table: SC_N nodes per turn, long double
    if (!g_sctab[dev]) {
        std::vector<double> tab(2 * 256);
        const long double two_pi = 6.283185307179586476925286766559005768L;
        for (int i = 0; i < 256; i++) { tab[2 * i] = (double)sinl(two_pi * i / 256); tab[2 * i + 1] = (double)cosl(two_pi * i / 256); }
        double* ptr = nullptr;
        hipError_t e = hipMalloc((void**)&ptr, tab.size() * sizeof(double));
        if (e == hipSuccess) e = hipMemcpy(ptr, tab.data(), tab.size() * sizeof(double), hipMemcpyHostToDevice);
        if (e != hipSuccess) { if (ptr) (void)hipFree(ptr); set_error("sincos table", e); return SIM5GPU_E_HIP; }
        g_sctab[dev] = ptr;
    }
    p.sctab = g_sctab[dev];
    return SIM5GPU_OK;
}

int attach_flux_table(DiskConsts& d)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return SIM5GPU_E_HIP;
    if (dev < 0 || dev >= FT_DEVICES) { snprintf(g_err, sizeof g_err, "device %d: more than %d devices", dev, FT_DEVICES); return SIM5GPU_E_ARG; }
    if (!t_pin_scope) { snprintf(g_err, sizeof g_err, "attach_flux_table outside a FluxPinScope"); return SIM5GPU_E_ARG; }
    std::vector<double*> to_free;
    // blocks nobody holds any more: freed after the lock has been left (every launch that read them was enqueued before its
    // entry point dropped the pin; the synchronisation waits for those launches)
    struct FreeLater {
        std::vector<double*>& v;
        ~FreeLater() { if (!v.empty()) { (void)hipDeviceSynchronize(); for (double* p : v) (void)hipFree(p); } }
    } free_later{to_free};
    std::lock_guard<std::mutex> hold(g_ftab_lock);
    const double wmin = d.x0 / 16.0;
    d.ft_wmin = wmin;
    d.ft_inv_dw = (double)s5abi::FT_N / (1.0 - wmin);
    FluxCache& Cc = g_ftab[dev];
    const FluxKey key = flux_key(d.a, d.scale);
    auto pin = [&](const double* ptr) { Cc.pinned[ptr]++; ((PinList*)t_pin_scope->held)->emplace_back(dev, ptr); };
    {
        auto it = Cc.live.find(key);
        if (it != Cc.live.end()) {
            it->second.stamp = ++Cc.clock;
            d.cold = it->second.ptr;
            d.ftab = it->second.usable ? it->second.ptr + s5abi::COLD_N : nullptr;
            pin(it->second.ptr);
            return SIM5GPU_OK;
        }
    }
    flux_cache_retire(Cc, to_free);
    std::vector<double> tab;
    const double fit_error = build_flux_table(d, wmin, tab);
    // towards a = 1 the inner edge approaches the logarithmic singularity at x1 and the uniform grid stops resolving
    // the profile (1e-10 at a = 0.998, 1e-8 at 0.9995): such spins keep the closed form (remembered as a block without table)
    const bool usable = (fit_error <= 2e-9);
    std::vector<double> block(s5abi::COLD_N, 0.0);
    const double cold[] = { d.a, d.x0, d.x1, d.x2, d.x3, d.p1, d.p2, d.p3, d.d1, d.d2, d.d3, d.mdot, d.mass };    // s5_disk.hpp ClosedFormConsts
    static_assert(sizeof cold / sizeof cold[0] <= s5abi::COLD_N, "cold block");
    for (size_t i = 0; i < sizeof cold / sizeof cold[0]; i++) block[i] = cold[i];
    if (usable) block.insert(block.end(), tab.begin(), tab.end());
    double* ptr = nullptr;
    hipError_t e = hipMalloc((void**)&ptr, block.size() * sizeof(double));
    if (e == hipSuccess) e = hipMemcpy(ptr, block.data(), block.size() * sizeof(double), hipMemcpyHostToDevice);
    if (e != hipSuccess) { if (ptr) (void)hipFree(ptr); set_error("flux table", e); return SIM5GPU_E_HIP; }
    Cc.live.emplace(key, FluxTable{ d.a, d.scale, ptr, usable, ++Cc.clock });
    d.cold = ptr;
    d.ftab = usable ? ptr + s5abi::COLD_N : nullptr;
    pin(ptr);
    return SIM5GPU_OK;
}

FluxPinScope::FluxPinScope() : held(new PinList), outer(t_pin_scope) { t_pin_scope = this; }
FluxPinScope::~FluxPinScope()
{
    PinList* mine = (PinList*)held;
    if (!mine->empty()) {
        std::lock_guard<std::mutex> hold(g_ftab_lock);
        for (auto& dp : *mine) {
            auto& pinned = g_ftab[dp.first].pinned;
            auto it = pinned.find(dp.second);
            if (it != pinned.end() && --it->second <= 0) pinned.erase(it);
        }
    }
    delete mine;
    t_pin_scope = (FluxPinScope*)outer;
}

// every flux-table block of every device given back (sim5gpu_release_workspaces); returns the bytes freed
size_t release_flux_tables()
{
    std::lock_guard<std::mutex> hold(g_ftab_lock);
    int cur = 0;
    (void)hipGetDevice(&cur);
    size_t freed = 0;
    for (int dev = 0; dev < FT_DEVICES; ++dev) {
        FluxCache& Cc = g_ftab[dev];
        if (Cc.live.empty() && Cc.resting.empty()) continue;
        (void)hipSetDevice(dev);
        (void)hipDeviceSynchronize();
        // (a block an entry point of another thread still holds stays: it goes to the resting list and is freed by a later retirement)
        std::vector<double*> keep;
        for (auto& kv : Cc.live) {
            if (Cc.pinned.count(kv.second.ptr)) { keep.push_back(kv.second.ptr); continue; }
            (void)hipFree(kv.second.ptr); freed += (size_t)(s5abi::COLD_N + (kv.second.usable ? s5abi::FT_N * (s5abi::FT_DEG + 1) : 0)) * sizeof(double);
        }
        for (double* p : Cc.resting) {
            if (Cc.pinned.count(p)) { keep.push_back(p); continue; }
            (void)hipFree(p); freed += (size_t)(s5abi::COLD_N + s5abi::FT_N * (s5abi::FT_DEG + 1)) * sizeof(double);   // (a resting block's size is not kept: counted as a full one)
        }
        Cc.live.clear(); Cc.resting.swap(keep);
    }
    (void)hipSetDevice(cur);
    return freed;
}

// validate a job description and turn it into the kernel argument block
// rows named by y0, y1 and the striping
static int image_rows_top(const sim5gpu_image_desc* desc)
{
    if (!desc || desc->y1 <= desc->y0) return 0;
    if (desc->stripe_rows <= 0) return desc->y1 - desc->y0;
    if (desc->stripe_step < desc->stripe_rows) return 0;       // overlapping or non-advancing stripes: rejected by the launchers
    int rows = 0;
    for (int y = desc->y0; y < desc->y1; y += desc->stripe_step)
        rows += (y + desc->stripe_rows <= desc->y1) ? desc->stripe_rows : desc->y1 - y;
    return rows;
}

int fill_image_params(const sim5gpu_image_desc* desc, ImageParams& p, bool need_disk)
{
    if (!desc) { snprintf(g_err, sizeof g_err, "image descriptor is NULL"); return SIM5GPU_E_ARG; }
    if (desc->nx <= 0 || desc->ny <= 0 || desc->y0 < 0 || desc->y1 > desc->ny || desc->y0 >= desc->y1) {
        snprintf(g_err, sizeof g_err, "bad image geometry nx=%d ny=%d rows=[%d,%d)", desc->nx, desc->ny,
                 desc->y0, desc->y1);
        return SIM5GPU_E_ARG;
    }
    if (desc->stripe_rows < 0 || (desc->stripe_rows > 0 && desc->stripe_step < desc->stripe_rows)) {
        snprintf(g_err, sizeof g_err, "bad striping stripe_rows=%d stripe_step=%d", desc->stripe_rows, desc->stripe_step);
        return SIM5GPU_E_ARG;
    }
    if ((desc->flags & SIM5GPU_IMG_MIRROR) && desc->y1 > (desc->ny + 1) / 2) {
        snprintf(g_err, sizeof g_err, "SIM5GPU_IMG_MIRROR: rows [%d,%d) must lie in the upper half of ny=%d", desc->y0, desc->y1, desc->ny);
        return SIM5GPU_E_ARG;
    }
    if (need_disk && (!(desc->bh_mass > 0.0) || !(desc->mdot > 0.0))) {     // a zero-initialised descriptor would give inf / NaN fluxes
        snprintf(g_err, sizeof g_err, "image descriptor needs bh_mass > 0 and mdot > 0 (got %g, %g)", desc->bh_mass, desc->mdot);
        return SIM5GPU_E_ARG;
    }
    memset(&p, 0, sizeof p);
    p.nx = desc->nx; p.ny = desc->ny; p.y0 = desc->y0; p.y1 = desc->y1;
    p.stripe_rows = desc->stripe_rows; p.stripe_step = desc->stripe_step;
    p.nrows = sim5gpu_image_rows(desc);
    p.mirror = (desc->flags & SIM5GPU_IMG_MIRROR) ? 1 : 0;
    p.inplace = (desc->flags & SIM5GPU_IMG_INPLACE) ? 1 : 0;
    p.direct = (desc->flags & SIM5GPU_IMG_DIRECT) ? 1 : 0;
    p.nrows_top = image_rows_top(desc);
    p.inv_nx = 1.0 / (double)desc->nx; p.inv_ny = 1.0 / (double)desc->ny;
    p.ny_over_nx = (double)desc->ny / (double)desc->nx;
    { const double ac = fmax(1e-4, desc->a); p.inv_2a2 = 1.0 / (2.0 * ac * ac); }
    p.max_order = desc->max_order > 0 ? desc->max_order : 2;
    p.a = desc->a;
    p.incl = desc->incl;
    reference_sincos(desc->incl, p.sin_i, p.cos_i);
    const double rms = host_r_ms(desc->a);                 // ref disk-image.c:41-42
    p.rms = desc->rms > 0.0 ? desc->rms : rms;
    p.rmax = desc->rmax > 0.0 ? desc->rmax : rms + 8.0;
    p.pol_degree = desc->pol_degree;
    p.disk = make_disk_consts(desc->bh_mass, desc->disk_spin >= 0.0 ? desc->disk_spin : desc->a, desc->mdot);   // ref disk-image.c:45
    return SIM5GPU_OK;
}

} // namespace s5

using namespace s5;

extern "C" {

int sim5gpu_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int sim5gpu_set_device(int device)
{
    if (!have_device()) return SIM5GPU_E_NO_DEVICE;
    S5_HIP(hipSetDevice(device));
    return SIM5GPU_OK;
}

const char* sim5gpu_last_error(void) { return g_err; }
const char* sim5gpu_version(void) { return "sim5_amd 0.1 (gfx950)"; }

int sim5gpu_malloc(void** dptr, size_t bytes)
{
    if (!dptr) return SIM5GPU_E_ARG;
    if (!have_device()) return SIM5GPU_E_NO_DEVICE;
    S5_HIP(hipMalloc(dptr, bytes ? bytes : 1));
    return SIM5GPU_OK;
}

int sim5gpu_free(void* dptr)
{
    if (!dptr) return SIM5GPU_OK;
    S5_HIP(hipFree(dptr));
    return SIM5GPU_OK;
}

int sim5gpu_memcpy_h2d(void* dst, const void* src, size_t bytes)
{
    if (!have_device()) return SIM5GPU_E_NO_DEVICE;
    S5_HIP(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
    return SIM5GPU_OK;
}

int sim5gpu_memcpy_d2h(void* dst, const void* src, size_t bytes)
{
    if (!have_device()) return SIM5GPU_E_NO_DEVICE;
    S5_HIP(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
    return SIM5GPU_OK;
}

int sim5gpu_memset(void* dst, int value, size_t bytes)
{
    if (!have_device()) return SIM5GPU_E_NO_DEVICE;
    S5_HIP(hipMemset(dst, value, bytes));
    return SIM5GPU_OK;
}

int sim5gpu_synchronize(void* stream)
{
    if (!have_device()) return SIM5GPU_E_NO_DEVICE;
    S5_HIP(hipStreamSynchronize((hipStream_t)stream));
    return SIM5GPU_OK;
}

int sim5gpu_event_create(void** event)
{
    if (!event) return SIM5GPU_E_ARG;
    if (!have_device()) return SIM5GPU_E_NO_DEVICE;
    hipEvent_t e;
    S5_HIP(hipEventCreate(&e));
    *event = (void*)e;
    return SIM5GPU_OK;
}

int sim5gpu_event_record(void* event, void* stream)
{
    if (!event) return SIM5GPU_E_ARG;
    S5_HIP(hipEventRecord((hipEvent_t)event, (hipStream_t)stream));
    return SIM5GPU_OK;
}

int sim5gpu_event_elapsed_ms(void* start, void* stop, float* ms)
{
    if (!start || !stop || !ms) return SIM5GPU_E_ARG;
    S5_HIP(hipEventSynchronize((hipEvent_t)stop));
    S5_HIP(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
    return SIM5GPU_OK;
}

int sim5gpu_event_destroy(void* event)
{
    if (!event) return SIM5GPU_OK;
    S5_HIP(hipEventDestroy((hipEvent_t)event));
    return SIM5GPU_OK;
}

// ---- disk model (process-global, like SIM5) -------------------------------------------------
static std::atomic<unsigned long> g_disk_generation{0ul};
int sim5gpu_disk_nt_setup(double M, double a, double mdot_or_L, double alpha, int options)
{
    if (options & ~SIM5GPU_DISK_NT_OPTION_LUMINOSITY) {
        snprintf(g_err, sizeof g_err, "disk_nt_setup: unknown option bits 0x%x", options);
        return SIM5GPU_E_ARG;
    }
    DiskConsts d = make_disk_consts(M, a, mdot_or_L, alpha);
    if (options & SIM5GPU_DISK_NT_OPTION_LUMINOSITY) {
        // the accretion rate whose integrated luminosity is mdot_or_L: the reference's bisection on [0, 100] to
        // 1e-6 (ref src/sim5disk-nt.c:371-385, src/sim5roots.c:21-63); every trial luminosity is the Simpson
        // integral of disk_lumi(), evaluated on the device
        if (!have_device()) return SIM5GPU_E_NO_DEVICE;
        const double L0 = mdot_or_L, x1 = 0.0, x2 = 100.0, xacc = 1e-6;
        double dx, f, fmid, xmid, rtb, L;
        int rc, j;
        disk_set_mdot(d, x2); if ((rc = disk_lumi(d, &L)) != 0) return rc; fmid = L0 - L;
        disk_set_mdot(d, x1); if ((rc = disk_lumi(d, &L)) != 0) return rc; f = L0 - L;
        if ((f * fmid) >= 0.0) {
            disk_set_mdot(d, 0.0);                       // not bracketed: the reference ends with mdot = 0
        } else {
            if (f < 0.0) { rtb = x1; dx = x2 - x1; } else { rtb = x2; dx = x1 - x2; }
            for (j = 0; j < 500; j++) {
                dx = dx * 0.5;
                xmid = rtb + dx;
                disk_set_mdot(d, xmid);
                if ((rc = disk_lumi(d, &L)) != 0) return rc;
                fmid = L0 - L;
                if (fmid <= 0.0) rtb = xmid;
                if ((fabs(dx) < xacc) || (fmid == 0.0)) break;
            }
            disk_set_mdot(d, (j >= 500) ? 0.0 : rtb);
        }
    }
    g_disk = d;
    g_disk_generation.fetch_add(1ul, std::memory_order_release);
    return SIM5GPU_OK;
}

// how many disk models this process has set up so far (0: none): whoever keeps values of disk_nt_* of the process-global
// model -- the host shim's per-ray and look-ahead records -- stamps them with it and compares before answering from them
unsigned long sim5gpu_disk_nt_generation(void) { return g_disk_generation.load(std::memory_order_acquire); }

int sim5gpu_disk_nt_mdot(double* mdot)
{
    if (!mdot) return SIM5GPU_E_ARG;
    if (!g_disk.ready) { snprintf(g_err, sizeof g_err, "disk_nt_setup has not been called"); return SIM5GPU_E_NOT_SETUP; }
    *mdot = g_disk.mdot;
    return SIM5GPU_OK;
}

int sim5gpu_disk_nt_lumi(double* lumi)
{
    if (!lumi) return SIM5GPU_E_ARG;
    if (!g_disk.ready) { snprintf(g_err, sizeof g_err, "disk_nt_setup has not been called"); return SIM5GPU_E_NOT_SETUP; }
    if (!have_device()) return SIM5GPU_E_NO_DEVICE;
    return disk_lumi(g_disk, lumi);
}

int sim5gpu_disk_nt_r_min(double* r_min)
{
    if (!r_min) return SIM5GPU_E_ARG;
    if (!g_disk.ready) { snprintf(g_err, sizeof g_err, "disk_nt_setup has not been called"); return SIM5GPU_E_NOT_SETUP; }
    // disk_nt_r_min() recomputes r0 + 1e-3 in double from the float spin (ref src/sim5disk-nt.c:91-105)
    const double a = g_disk.a;
    const double sga = (a >= 0.0) ? +1. : -1.;
    const double z1 = 1. + pow(1. - a * a, 1. / 3.) * (pow(1. + a, 1. / 3.) + pow(1. - a, 1. / 3.));
    const double z2 = sqrt(3. * a * a + z1 * z1);
    *r_min = 3. + z2 - sga * sqrt((3. - z1) * (3. + z1 + 2. * z2)) + 1e-3;
    return SIM5GPU_OK;
}

// number of (packed) output rows of a job description: y1 - y0, or the total height of its stripes
int sim5gpu_image_rows(const sim5gpu_image_desc* desc)
{
    const int top = image_rows_top(desc);
    if (top <= 0 || !(desc->flags & SIM5GPU_IMG_MIRROR)) return top;
    if (desc->y1 > (desc->ny + 1) / 2) return 0;               // mirrored rows must be named in the upper half: rejected
    // the middle row of an odd image is its own mirror: it is the last row named, if it is named at all
    bool has_middle = false;
    if (desc->ny % 2 == 1) {
        const int mid = (desc->ny - 1) / 2;
        if (desc->stripe_rows <= 0) has_middle = (mid >= desc->y0 && mid < desc->y1);
        else if (mid >= desc->y0 && mid < desc->y1) has_middle = ((mid - desc->y0) % desc->stripe_step) < desc->stripe_rows;
    }
    return 2 * top - (has_middle ? 1 : 0);
}

// ---- whole-job image entry points --------------------------------------------------------------
static void attach_aux(ImageParams& p, const sim5gpu_image_aux* aux)
{
    if (!aux) return;
    p.cls = aux->cls; p.gtype = aux->gtype; p.r = aux->r; p.g = aux->g; p.flux = aux->flux;
}

int sim5gpu_disk_image(const sim5gpu_image_desc* desc, float* d_image_f, float* d_image_g,
                       const sim5gpu_image_aux* d_aux, void* stream)
{
    if (!d_image_f || !d_image_g) { snprintf(g_err, sizeof g_err, "output image pointers are NULL"); return SIM5GPU_E_ARG; }
    ImageParams p;
    int rc = fill_image_params(desc, p);
    if (rc) return rc;
    if (!have_device()) return SIM5GPU_E_NO_DEVICE;
    FluxPinScope pins;                                       // the table block stays until the launch below has been enqueued
    if (!(desc->flags & SIM5GPU_IMG_STRICT) && ((rc = attach_flux_table(p.disk)) != 0 || (rc = attach_K_table(p)) != 0)) return rc;
    p.img_f = d_image_f; p.img_g = d_image_g;
    attach_aux(p, d_aux);
    hipError_t e = (hipError_t)((desc->flags & SIM5GPU_IMG_STRICT) ? s5_launch_disk_image_strict(p, (hipStream_t)stream)
                                                                 : s5_launch_disk_image_fast(p, (hipStream_t)stream));
    if (e != hipSuccess) { set_error("disk_image launch", e); return SIM5GPU_E_HIP; }
    return SIM5GPU_OK;
}

/* Several image jobs, ONE launch where the jobs allow it: jobs of the default (fast) variant whose row set is symmetric about
 * the middle of the image -- whole images, centred bands, SIM5GPU_IMG_MIRROR shares -- go through the job-list kernel
 * (k_disk_image.hip: disk_image_jobs_kernel) in groups of up to 16, the jobs of a group streaming through the GPU back to
 * back; any other job (strict variant, an asymmetric row range) is launched by itself, in order.  Images are the bits
 * sim5gpu_disk_image gives job by job. */
int sim5gpu_disk_image_jobs(int n_jobs, const sim5gpu_image_desc* descs, float* const* d_image_f, float* const* d_image_g, void* stream)
{
    if (n_jobs == 0) return SIM5GPU_OK;
    if (n_jobs < 0 || !descs || !d_image_f || !d_image_g) { snprintf(g_err, sizeof g_err, "disk_image_jobs: bad arguments"); return SIM5GPU_E_ARG; }
    // validate everything before anything is launched
    std::vector<ImageParams> ps((size_t)n_jobs);
    for (int j = 0; j < n_jobs; ++j) {
        if (!d_image_f[j] || !d_image_g[j]) { snprintf(g_err, sizeof g_err, "disk_image_jobs: job %d: output image pointers are NULL", j); return SIM5GPU_E_ARG; }
        const int rc = fill_image_params(&descs[j], ps[(size_t)j]);
        if (rc) return rc;
        ps[(size_t)j].img_f = d_image_f[j]; ps[(size_t)j].img_g = d_image_g[j];
    }
    if (!have_device()) return SIM5GPU_E_NO_DEVICE;
    FluxPinScope pins;                                       // every job's table block stays until the last launch has been enqueued
    std::vector<ImageParams> group;
    auto flush = [&]() -> int {
        if (group.empty()) return SIM5GPU_OK;
        hipError_t e = (hipError_t)s5_launch_disk_image_jobs_fast(group.data(), (int)group.size(), (hipStream_t)stream);
        group.clear();
        if (e != hipSuccess) { set_error("disk_image_jobs launch", e); return SIM5GPU_E_HIP; }
        return SIM5GPU_OK;
    };
    for (int j = 0; j < n_jobs; ++j) {
        ImageParams& p = ps[(size_t)j];
        const bool strict = (descs[j].flags & SIM5GPU_IMG_STRICT) != 0;
        int rc;
        if (!strict && ((rc = attach_flux_table(p.disk)) != 0 || (rc = attach_K_table(p)) != 0)) return rc;
        if (!strict && s5_jobs_eligible(p)) {
            group.push_back(p);
            if ((int)group.size() == s5abi::JOBS_MAX && (rc = flush()) != 0) return rc;
            continue;
        }
        if ((rc = flush()) != 0) return rc;                  // keeps the order of the jobs on the stream
        hipError_t e = (hipError_t)(strict ? s5_launch_disk_image_strict(p, (hipStream_t)stream) : s5_launch_disk_image_fast(p, (hipStream_t)stream));
        if (e != hipSuccess) { set_error("disk_image_jobs launch", e); return SIM5GPU_E_HIP; }
    }
    return flush();
}

int sim5gpu_disk_rays(const sim5gpu_image_desc* desc, size_t n, const double* d_alpha,
                      const double* d_beta, float* d_image_f, float* d_image_g,
                      const sim5gpu_image_aux* d_aux, void* stream)
{
    if (!d_image_f || !d_image_g || !d_alpha || !d_beta) {
        snprintf(g_err, sizeof g_err, "disk_rays: NULL pointer argument");
        return SIM5GPU_E_ARG;
    }
    sim5gpu_image_desc dd;
    if (!desc) return SIM5GPU_E_ARG;
    dd = *desc;
    if (dd.nx <= 0) { dd.nx = 1; dd.ny = 1; dd.y0 = 0; dd.y1 = 1; }   // geometry unused in list mode
    ImageParams p;
    int rc = fill_image_params(&dd, p);
    if (rc) return rc;
    if (n == 0) return SIM5GPU_OK;
    if (desc->flags & SIM5GPU_IMG_INPLACE) { snprintf(g_err, sizeof g_err, "disk_rays: SIM5GPU_IMG_INPLACE needs the pixel grid"); return SIM5GPU_E_ARG; }
    if (!have_device()) return SIM5GPU_E_NO_DEVICE;
    FluxPinScope pins;                                       // the table block stays until the launch below has been enqueued
    if (!(desc->flags & SIM5GPU_IMG_STRICT) && ((rc = attach_flux_table(p.disk)) != 0 || (rc = attach_K_table(p)) != 0)) return rc;
    p.img_f = d_image_f; p.img_g = d_image_g;
    p.alpha = d_alpha; p.beta = d_beta; p.n = n;
    attach_aux(p, d_aux);
    hipError_t e = (hipError_t)((desc->flags & SIM5GPU_IMG_STRICT) ? s5_launch_disk_image_strict(p, (hipStream_t)stream)
                                                                 : s5_launch_disk_image_fast(p, (hipStream_t)stream));
    if (e != hipSuccess) { set_error("disk_rays launch", e); return SIM5GPU_E_HIP; }
    return SIM5GPU_OK;
}

int sim5gpu_disk_image_host(const sim5gpu_image_desc* desc, float* h_image_f, float* h_image_g,
                            const sim5gpu_image_aux* h_aux)
{
    if (!h_image_f || !h_image_g) return SIM5GPU_E_ARG;
    ImageParams chk;
    int rc = fill_image_params(desc, chk);
    if (rc) return rc;
    if (desc->flags & SIM5GPU_IMG_INPLACE) { snprintf(g_err, sizeof g_err, "disk_image_host: SIM5GPU_IMG_INPLACE is for device buffers"); return SIM5GPU_E_ARG; }
    if (!have_device()) return SIM5GPU_E_NO_DEVICE;
    const size_t n = (size_t)sim5gpu_image_rows(desc) * (size_t)desc->nx;
    DevBuf<float> f(n), g(n);
    DevBuf<uint8_t> cls(h_aux && h_aux->cls ? n : 0);
    DevBuf<int8_t> gt(h_aux && h_aux->gtype ? n : 0);
    DevBuf<double> r(h_aux && h_aux->r ? n : 0), gg(h_aux && h_aux->g ? n : 0), fl(h_aux && h_aux->flux ? n : 0);
    if (!f.ok() || !g.ok() || !cls.ok() || !gt.ok() || !r.ok() || !gg.ok() || !fl.ok()) {
        snprintf(g_err, sizeof g_err, "disk_image_host: device allocation failed");
        return SIM5GPU_E_HIP;
    }
    sim5gpu_image_aux daux = { cls.ptr, gt.ptr, r.ptr, gg.ptr, fl.ptr };
    rc = sim5gpu_disk_image(desc, f.ptr, g.ptr, &daux, nullptr);
    if (rc) return rc;
    S5_HIP(hipDeviceSynchronize());
    S5_HIP(f.to_host(h_image_f));
    S5_HIP(g.to_host(h_image_g));
    if (h_aux) {
        if (h_aux->cls) S5_HIP(cls.to_host(h_aux->cls));
        if (h_aux->gtype) S5_HIP(gt.to_host(h_aux->gtype));
        if (h_aux->r) S5_HIP(r.to_host(h_aux->r));
        if (h_aux->g) S5_HIP(gg.to_host(h_aux->g));
        if (h_aux->flux) S5_HIP(fl.to_host(h_aux->flux));
    }
    return SIM5GPU_OK;
}

/* validation of a job description as the image launchers apply it, host arithmetic only (no GPU): 0 or SIM5GPU_E_ARG */
int sim5gpu_image_desc_check(const sim5gpu_image_desc* desc)
{
    ImageParams p;
    const int rc = fill_image_params(desc, p, true);
    if (rc) return rc;
    if (p.nrows <= 0) { snprintf(g_err, sizeof g_err, "image description names no rows"); return SIM5GPU_E_ARG; }
    return SIM5GPU_OK;
}

/* PCI bus id of a device ("0000:05:00.0"), so that a multi-process job can show that its ranks sit on distinct GPUs */
int sim5gpu_image_view(const sim5gpu_image_desc* desc, double* rmax, double* rms, double* sin_i, double* cos_i)
{
    if (!desc || !rmax || !rms) { snprintf(g_err, sizeof g_err, "image_view: NULL pointer argument"); return SIM5GPU_E_ARG; }
    const double r = host_r_ms(desc->a);
    *rms = desc->rms > 0.0 ? desc->rms : r;
    *rmax = desc->rmax > 0.0 ? desc->rmax : r + 8.0;
    double s, c;
    reference_sincos(desc->incl, s, c);
    if (sin_i) *sin_i = s;
    if (cos_i) *cos_i = c;
    return SIM5GPU_OK;
}

int sim5gpu_device_bus_id(int device, char* buf, int len)
{
    if (!buf || len < 16) return SIM5GPU_E_ARG;
    if (!have_device()) return SIM5GPU_E_NO_DEVICE;
    S5_HIP(hipDeviceGetPCIBusId(buf, len, device));
    return SIM5GPU_OK;
}

/* image row of every packed output row of a job description (host arithmetic, no GPU): rows[i] for i < min(count, capacity) */
int sim5gpu_image_row_map(const sim5gpu_image_desc* desc, int* rows, int capacity)
{
    ImageParams p;
    const int rc = fill_image_params(desc, p, false);
    if (rc) return rc;
    if (!rows && capacity > 0) return SIM5GPU_E_ARG;
    for (int lr = 0; lr < p.nrows && lr < capacity; ++lr) rows[lr] = image_row(p, lr);
    return SIM5GPU_OK;
}

int sim5gpu_image_place_shares(int n_shares, const sim5gpu_image_desc* descs, const float* d_shares, size_t share_rows,
                               float* d_image_f, float* d_image_g, void* stream)
{
    if (n_shares == 0) return SIM5GPU_OK;
    if (n_shares < 0 || n_shares > 16 || !descs || !d_shares || !d_image_f || !d_image_g) {
        snprintf(g_err, sizeof g_err, "image_place_shares: bad arguments (1..16 shares, non-NULL pointers)");
        return SIM5GPU_E_ARG;
    }
    RowMap maps[16];
    for (int i = 0; i < n_shares; ++i) {
        ImageParams p;
        const int rc = fill_image_params(&descs[i], p, false);
        if (rc) return rc;
        if (p.nx != descs[0].nx || p.ny != descs[0].ny || (size_t)p.nrows > share_rows) {
            snprintf(g_err, sizeof g_err, "image_place_shares: share %d does not fit (nx=%d ny=%d rows=%d, block rows=%zu)", i, p.nx, p.ny, p.nrows, share_rows);
            return SIM5GPU_E_ARG;
        }
        maps[i] = RowMap{ p.ny, p.y0, p.nrows, p.nrows_top, p.stripe_rows, p.stripe_step, p.mirror };
    }
    if (!have_device()) return SIM5GPU_E_NO_DEVICE;
    hipError_t e = (hipError_t)s5_launch_place_shares(n_shares, maps, d_shares, share_rows, descs[0].nx, d_image_f, d_image_g, (hipStream_t)stream);
    if (e != hipSuccess) { set_error("image_place_shares launch", e); return SIM5GPU_E_HIP; }
    return SIM5GPU_OK;
}

/* peer-to-peer exchange: the root's image mapped into the peers' address spaces (include/sim5gpu.h) */
int sim5gpu_ipc_export(const void* d_ptr, void* handle)
{
    static_assert(sizeof(hipIpcMemHandle_t) <= SIM5GPU_IPC_HANDLE_BYTES, "IPC handle size");
    if (!d_ptr || !handle) { snprintf(g_err, sizeof g_err, "ipc_export: NULL pointer argument"); return SIM5GPU_E_ARG; }
    if (!have_device()) return SIM5GPU_E_NO_DEVICE;
    hipIpcMemHandle_t h;
    S5_HIP(hipIpcGetMemHandle(&h, const_cast<void*>(d_ptr)));
    memset(handle, 0, SIM5GPU_IPC_HANDLE_BYTES);
    memcpy(handle, &h, sizeof h);
    return SIM5GPU_OK;
}

int sim5gpu_ipc_open(const void* handle, void** d_ptr)
{
    if (!handle || !d_ptr) { snprintf(g_err, sizeof g_err, "ipc_open: NULL pointer argument"); return SIM5GPU_E_ARG; }
    if (!have_device()) return SIM5GPU_E_NO_DEVICE;
    hipIpcMemHandle_t h;
    memcpy(&h, handle, sizeof h);
    *d_ptr = nullptr;
    S5_HIP(hipIpcOpenMemHandle(d_ptr, h, hipIpcMemLazyEnablePeerAccess));
    return SIM5GPU_OK;
}

int sim5gpu_ipc_close(void* d_ptr)
{
    if (!d_ptr) return SIM5GPU_OK;
    if (!have_device()) return SIM5GPU_E_NO_DEVICE;
    S5_HIP(hipIpcCloseMemHandle(d_ptr));
    return SIM5GPU_OK;
}

/* self-check utility: number of 32-bit words in which two DEVICE buffers differ (synchronous; bit comparison) */
int sim5gpu_words_differ(const void* d_a, const void* d_b, size_t n_words, unsigned long long* h_count)
{
    if (!h_count || ((!d_a || !d_b) && n_words)) { snprintf(g_err, sizeof g_err, "words_differ: NULL pointer argument"); return SIM5GPU_E_ARG; }
    *h_count = 0;
    if (n_words == 0) return SIM5GPU_OK;
    if ((((size_t)d_a | (size_t)d_b) & 3) != 0) { snprintf(g_err, sizeof g_err, "words_differ: buffers must be 4-byte aligned"); return SIM5GPU_E_ARG; }
    if (!have_device()) return SIM5GPU_E_NO_DEVICE;
    DevBuf<unsigned long long> cnt(1);
    if (!cnt.ok() || !cnt.ptr) { snprintf(g_err, sizeof g_err, "words_differ: device allocation failed"); return SIM5GPU_E_HIP; }
    if (cnt.pinned) *cnt.ptr = 0; else S5_HIP(hipMemset(cnt.ptr, 0, sizeof(unsigned long long)));
    hipError_t e = (hipError_t)s5_launch_words_differ(d_a, d_b, n_words, cnt.ptr, nullptr);
    if (e != hipSuccess) { set_error("words_differ launch", e); return SIM5GPU_E_HIP; }
    S5_HIP(hipStreamSynchronize(nullptr));
    S5_HIP(cnt.to_host(h_count));
    return SIM5GPU_OK;
}

} // extern "C"
