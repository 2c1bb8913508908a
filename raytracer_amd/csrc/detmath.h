// detmath.h -- deterministic double-precision sin/cos/pow5 built from IEEE +,-,* only.
//
// Why: the reference draws directions through libm cos/sin/pow (src/sppm2.h:140-152,276,498;
// src/radiance.h:39-42).  glibc and the ROCm device library differ in the last ulp, so a GPU
// cannot be bit-identical to a glibc build.  These three functions use nothing
// but correctly rounded IEEE-754 add/sub/mul (no FMA: every translation unit that includes this
// file is built with -ffp-contract=off), so the same source gives the same bits on x86-64 and on
// gfx950.  Accuracy is about 1 ulp on the ranges the path uses (x in [0, 2*pi], c in [0, 1]),
// which tests/test_detmath.py checks against libm.
//
// Algorithm: Cody-Waite reduction by k*(pi/2) with a two-part pi/2, then the classic minimax
// kernels for sin and cos on [-pi/4, pi/4] (published fdlibm coefficients, Sun Microsystems 1993).
#ifndef SPPM_DETMATH_H
#define SPPM_DETMATH_H

#if defined(__HIPCC__)
#define DM_HD __host__ __device__ inline
#else
#define DM_HD static inline
#endif

DM_HD double dm_ksin(double r)
{
    const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
                 S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
                 S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
    const double z = r * r;
    const double p = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
    return r + (r * z) * (S1 + z * p);
}

DM_HD double dm_kcos(double r)
{
    const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
                 C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
                 C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
    const double z = r * r;
    const double p = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
    const double hz = 0.5 * z;
    const double w = 1.0 - hz;
    return w + (((1.0 - w) - hz) + z * p);
}

// sin and cos of x for 0 <= x <= ~6.3 (the path only ever passes 2*pi*u, u in [0,1]).
DM_HD void dm_sincos(double x, double* s, double* c)
{
    const double INV_PIO2 = 6.36619772367581382433e-01;
    const double PIO2_HI = 1.57079632673412561417e+00;   // first 33 bits of pi/2
    const double PIO2_LO = 6.07710050650619224932e-11;   // pi/2 - PIO2_HI
    const int k = (int)(x * INV_PIO2 + 0.5);
    const double fk = (double)k;
    const double r = (x - fk * PIO2_HI) - fk * PIO2_LO;
    const double sr = dm_ksin(r), cr = dm_kcos(r);
    switch (k & 3) {
        case 0: *s = sr; *c = cr; break;
        case 1: *s = cr; *c = -sr; break;
        case 2: *s = -sr; *c = -cr; break;
        default: *s = -cr; *c = sr; break;
    }
}

DM_HD double dm_pow5(double c)
{
    const double c2 = c * c;
    const double c4 = c2 * c2;
    return c4 * c;
}

/* x / pi, correctly rounded, without the hardware division sequence (the gather divides three flux
 * components by pi for every accepted photon).  Markstein's scheme: y = RN(1/pi); a first quotient, two
 * exact-residual corrections with fused multiply-adds.  The last step returns RN(x/pi) whenever the
 * quotient before it is within one ulp, which the first correction ensures; x = 0 gives 0.  Valid for
 * finite x whose quotient and residuals stay in the normal range (fluxes do).  tests/test_host.py
 * compares it with the division on random and on hard (near-midpoint) arguments. */
DM_HD double dm_div_pi(double x)
{
    const double pi = 3.14159265358979323846;
    const double y = 0.31830988618379067154;  /* RN(1/pi) = 0x1.45f306dc9c883p-2 */
    double q = x * y;
    double r = __builtin_fma(-q, pi, x);
    q = __builtin_fma(r, y, q);
    r = __builtin_fma(-q, pi, x);
    return __builtin_fma(r, y, q);
}

#endif
