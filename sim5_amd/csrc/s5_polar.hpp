// s5_polar.hpp -- Walker-Penrose polarization transport and black-body radiance, gfx950 device code.
// Restated from the reference (ref: /root/reference/src/sim5polarization.c:145-158 constant,
// :55-105 vector, :249-258 constant at infinity, :272-285 angle rotation;
// src/sim5radiation.c:27-49 blackbody_Iv with the CGS constants of src/sim5const.h:30-41,86-87).
#pragma once
#include "s5_kerr.hpp"

namespace S5NS {

S5_DEV void polarization_constant(const double k[4], const double f[4], const Metric& g, double wp[2])
{
    const double a = g.a, m = g.m, r = g.r;
    const double A1 = (k[0] * f[1] - k[1] * f[0]) + a * (1. - m * m) * (k[1] * f[3] - k[3] * f[1]);
    const double A2 = msqrt(1. - m * m) * ((r * r + a * a) * (k[3] * f[2] - k[2] * f[3]) - a * (k[0] * f[2] - k[2] * f[0]));
    wp[0] = +r * A1 - a * m * A2;
    wp[1] = -r * A2 - a * m * A1;
}

S5_DEV void polarization_vector(const double k[4], const double wp[2], const Metric& g, double f[4])
{
    const double a = g.a, r = g.r;
    double m = g.m;
    double s = msqrt(1.0 - m * m);
    const double ra2 = r * r + a * a;
    const double r2 = r * r;
    const double a2 = a * a;
    double s2 = 1.0 - m * m;
    if (s < 1e-12) { s = 1e-12; s2 = 1e-24; m = 1.0 - 0.5 * s; }
    const double A1 = mdiv(+r * wp[0] - a * m * wp[1], r * r + a * a * m * m);
    const double A2 = mdiv(-r * wp[1] - a * m * wp[0], r * r + a * a * m * m);
    f[0] = 0.0;
    f[3] = mdiv(
             + g.g11 * A1 * k[1] * (s * r2 * k[3] + s * a2 * k[3] - s * a * k[0])
             + g.g22 * A2 * k[2] * (k[0] - a * s2 * k[3])
           , (
             + sq(k[0]) * g.g33 * (s * k[3] * a)
             + sq(k[0]) * g.g03 * (s * k[0] * a - s * r2 * k[3] - s * a2 * k[3] - a2 * s * s2 * k[3])
             + sq(k[1]) * g.g11 * a * s * s2 * (+r2 * k[3] + a2 * k[3] - a * k[0])
             + sq(k[2]) * g.g22 * (a2 * a * s * s2 * k[3] + r2 * a * s * s2 * k[3] - s * r2 * k[0] - s * a2 * k[0])
             + sq(k[3]) * g.g33 * s * (k[3] * a * s2 * r2 + k[3] * a2 * a * s2 - k[0] * r2 - k[0] * a2 - a2 * s2 * k[0])
             + sq(k[3]) * g.g03 * a * s * s2 * (r2 * k[0] + a2 * k[0])
           ));
    f[1] = mdiv(A1 - a * s * s * k[1] * f[3], k[0] - a * s * s * k[3]);
    f[2] = mdiv(A2 + s * k[2] * f[3] * ra2, s * k[3] * ra2 - s * a * k[0]);
    normalize_to(f, 1.0, g);
}

// sin_i = sin(inclination), supplied by the caller
S5_DEV void polarization_constant_infinity(double a, double alpha, double beta, double sin_i, double wp[2])
{
    const double gamma = -alpha - a * sin_i;
    wp[0] = -gamma;
    wp[1] = -beta;
}

S5_DEV double polarization_angle_rotation(double a, double sin_i, double alpha, double beta, const double wp[2])
{
    const double S = -alpha - a * sin_i;
    const double T = +beta;
    const double X = mdiv(-S * wp[1] - T * wp[0], S * S + T * T);
    const double Y = mdiv(-S * wp[0] + T * wp[1], S * S + T * T);
    return matan2(Y, X);
}

S5_DEV double blackbody_Iv(double T, double hardf, double cos_mu, double E)
{
    const double h = 6.626069e-27, c = 2.997925e+10, kB = 1.380650e-16;
    const double kev2freq = 2.417990e+17, freq2kev = 4.135667e-18;
    if (T <= 0.0) return 0.0;
    const double limbf = (cos_mu >= 0.0) ? 0.5 + 0.75 * cos_mu : 1.0;
    const double freq = kev2freq * E;
    return limbf * 2.0 * h * (freq * freq * freq) / sq(c) / (hardf * hardf * hardf * hardf) /
           expm1((h * freq) / (kB * hardf * T)) * (1. / freq2kev);
}

} // namespace S5NS
