// se3_math.hpp -- the O(1) mathematics of one align() iteration, written once
// for host and device (the device-resident loop runs it in k_post_flow /
// k_post_step; the C-ABI exports the same functions for host callers).
//
//   inverse transform         ref src/cvo.cpp:83-87
//   kernel thresholds         ref src/cvo.cpp:102-103
//   step-size cubic           ref src/cvo.cpp:53-69,291-307
//   Exp_SEK3 (K = 1)          ref src/LieGroup.cpp:159-186
//   dist_se3                  ref src/cvo.cpp:71-81
//   length-scale schedules    ref src/cvo.cpp:408-410, src/adaptive_cvo.cpp:538-545
//
// Arithmetic contract: float32 expressions in Eigen's coefficient
// order with NO contraction (-ffp-contract=off on host and device); anything
// transcendental is computed in float64 from +,-,*,/,sqrt,floor only (so that
// CPU and GPU agree bit for bit) and rounded once to float32.
#pragma once

#include <hip/hip_runtime.h>

#include <math.h>
#include <string.h>

#define CVO_HD __host__ __device__ inline __attribute__((always_inline))

namespace cvo_math {

struct Mat3 {
    float m[9];   // row-major
};

CVO_HD float at(const Mat3 &a, int r, int c) { return a.m[3 * r + c]; }

CVO_HD Mat3 identity3()
{
    Mat3 I;
    for (int k = 0; k < 9; ++k) I.m[k] = 0.0f;
    I.m[0] = I.m[4] = I.m[8] = 1.0f;
    return I;
}

CVO_HD Mat3 mul(const Mat3 &a, const Mat3 &b)
{
    Mat3 o;
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c)
            o.m[3 * r + c] = (at(a, r, 0) * at(b, 0, c) + at(a, r, 1) * at(b, 1, c)) + at(a, r, 2) * at(b, 2, c);
    return o;
}

CVO_HD void mulv(const Mat3 &a, const float x[3], float out[3])
{
    float t[3];
    for (int r = 0; r < 3; ++r) t[r] = (at(a, r, 0) * x[0] + at(a, r, 1) * x[1]) + at(a, r, 2) * x[2];
    for (int r = 0; r < 3; ++r) out[r] = t[r];
}

CVO_HD Mat3 skew(const float w[3])
{ // ref src/LieGroup.cpp:20-27
    Mat3 M;
    M.m[0] = 0.0f;  M.m[1] = -w[2]; M.m[2] = w[1];
    M.m[3] = w[2];  M.m[4] = 0.0f;  M.m[5] = -w[0];
    M.m[6] = -w[1]; M.m[7] = w[0];  M.m[8] = 0.0f;
    return M;
}

// Vector3f::squaredNorm() (fixed size 3, non-vectorised unrolled redux)
CVO_HD float sqnorm_fixed3(const float a[3]) { return a[0] * a[0] + (a[1] * a[1] + a[2] * a[2]); }
CVO_HD float norm_fixed3(const float a[3]) { return sqrtf(sqnorm_fixed3(a)); }

// [Rt | t] = [R^T | -R^T T]
CVO_HD void inverse_tf(const float R[9], const float T[3], float Rt[9], float t[3])
{
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) Rt[3 * r + c] = R[3 * c + r];
    for (int r = 0; r < 3; ++r)
        t[r] = ((-Rt[3 * r]) * T[0] + (-Rt[3 * r + 1]) * T[1]) + (-Rt[3 * r + 2]) * T[2];
}

CVO_HD void tf_to_mat4(const float Rt[9], const float t[3], float m[16])
{
    for (int k = 0; k < 16; ++k) m[k] = 0.0f;
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) m[4 * r + c] = Rt[3 * r + c];
        m[4 * r + 3] = t[r];
    }
    m[15] = 1.0f;
}

CVO_HD void mat4_mul(const float a[16], const float b[16], float out[16])
{
    float t[16];
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c)
            t[4 * r + c] = ((a[4 * r] * b[c] + a[4 * r + 1] * b[4 + c]) + a[4 * r + 2] * b[8 + c]) +
                           a[4 * r + 3] * b[12 + c];
    for (int k = 0; k < 16; ++k) out[k] = t[k];
}

// ---------------------------------------------------------------------------
// sin / cos in float64 from basic operations only (bit-reproducible on CPU
// and GPU): Cody-Waite reduction by pi/2, Taylor polynomials on [-pi/4, pi/4].
// Absolute error < 1e-16 for |x| < 1e5; callers round the result to float32.
// ---------------------------------------------------------------------------
CVO_HD void sincos_det(double x, double *s_out, double *c_out)
{
    const double two_over_pi = 0.63661977236758134308;
    const double pio2_hi = 1.57079632673412561417e+00;   // first 33 bits of pi/2
    const double pio2_lo = 6.07710050650619224932e-11;   // pi/2 - pio2_hi
    const double kd = floor(x * two_over_pi + 0.5);
    const double r = (x - kd * pio2_hi) - kd * pio2_lo;
    const double r2 = r * r;
    // sin r = r (1 - r2/6 (1 - r2/20 (1 - r2/42 (...))))  nested Taylor, degree 19
    double ps = 1.0;
    ps = 1.0 - r2 * (1.0 / (18.0 * 19.0)) * ps;
    ps = 1.0 - r2 * (1.0 / (16.0 * 17.0)) * ps;
    ps = 1.0 - r2 * (1.0 / (14.0 * 15.0)) * ps;
    ps = 1.0 - r2 * (1.0 / (12.0 * 13.0)) * ps;
    ps = 1.0 - r2 * (1.0 / (10.0 * 11.0)) * ps;
    ps = 1.0 - r2 * (1.0 / (8.0 * 9.0)) * ps;
    ps = 1.0 - r2 * (1.0 / (6.0 * 7.0)) * ps;
    ps = 1.0 - r2 * (1.0 / (4.0 * 5.0)) * ps;
    ps = 1.0 - r2 * (1.0 / (2.0 * 3.0)) * ps;
    const double sr = r * ps;
    double pc = 1.0;
    pc = 1.0 - r2 * (1.0 / (17.0 * 18.0)) * pc;
    pc = 1.0 - r2 * (1.0 / (15.0 * 16.0)) * pc;
    pc = 1.0 - r2 * (1.0 / (13.0 * 14.0)) * pc;
    pc = 1.0 - r2 * (1.0 / (11.0 * 12.0)) * pc;
    pc = 1.0 - r2 * (1.0 / (9.0 * 10.0)) * pc;
    pc = 1.0 - r2 * (1.0 / (7.0 * 8.0)) * pc;
    pc = 1.0 - r2 * (1.0 / (5.0 * 6.0)) * pc;
    pc = 1.0 - r2 * (1.0 / (3.0 * 4.0)) * pc;
    pc = 1.0 - r2 * (1.0 / (1.0 * 2.0)) * pc;
    const double cr = pc;
    // quadrant = kd mod 4 (kd may be negative)
    const double q4 = kd - 4.0 * floor(kd * 0.25);
    double s, c;
    if (q4 == 0.0) { s = sr; c = cr; }
    else if (q4 == 1.0) { s = cr; c = -sr; }
    else if (q4 == 2.0) { s = -sr; c = -cr; }
    else { s = -cr; c = sr; }
    *s_out = s;
    *c_out = c;
}

// ---------------------------------------------------------------------------
// Smallest positive real root of 4E s^3 + 3D s^2 + 2C s + B, else min_step;
// clamp to 0.8.  The float coefficients and the float division by the leading
// one are the reference's (VectorXf p_coef, companion-matrix first row); the
// roots themselves are bracketed between the stationary points of the monic
// cubic and refined in float64 until their float32 value is decided (the
// reference runs a float QR eigen-solve and accepts eigenvalues with imag()==0).
// ---------------------------------------------------------------------------
CVO_HD double cubic_eval(double a, double b, double c, double s) { return ((s + a) * s + b) * s + c; }

// x / 3.0, correctly rounded, without the division (~30 dependent instructions on the one chain of the
// post kernels): q0 = x * fl(1/3), r = fma(-3, q0, x) is exact, q = fma(r, fl(1/3), q0).  x / 3 is never
// within 1/6 ulp of a rounding boundary (x is a whole number of ulps, a boundary lies at a half), and q
// misses the exact quotient by far less: the same double as the division, bit for bit
// (tests/test_host_math.py holds it against x / 3.0 over the exponent range).
CVO_HD double div3(double x)
{
    const double c = 1.0 / 3.0;
    const double q0 = x * c;
    const double r = __builtin_fma(-3.0, q0, x);
    return __builtin_fma(r, c, q0);
}

// A bracket [lo,hi] of the monic cubic s^3 + a s^2 + b s + c that holds its
// smallest positive real root (sign change in the stated direction), if any.
struct CubicBracket {
    double a, b, c, lo, hi;
    bool found, increasing;
};

CVO_HD CubicBracket cubic_bracket(const double bcde[4])
{
    CubicBracket B;
    B.a = B.b = B.c = B.lo = B.hi = 0.0;
    B.found = false;
    B.increasing = true;
    const float c3 = (float)(4.0 * (float)bcde[3]);
    const float c2 = (float)(3.0 * (float)bcde[2]);
    const float c1 = (float)(2.0 * (float)bcde[1]);
    const float c0 = (float)bcde[0];
    const bool finite = (c3 == c3) && (c2 == c2) && (c1 == c1) && (c0 == c0) &&
                        fabsf(c3) <= 3.0e38f && fabsf(c2) <= 3.0e38f && fabsf(c1) <= 3.0e38f &&
                        fabsf(c0) <= 3.0e38f;
    if (!(c3 != 0.0f && finite)) return B;
    const float qa = c2 / c3, qb = c1 / c3, qc = c0 / c3;
    if (!(fabsf(qa) <= 3.0e38f && fabsf(qb) <= 3.0e38f && fabsf(qc) <= 3.0e38f)) return B;
    const double a = (double)qa, b = (double)qb, c = (double)qc;
    B.a = a; B.b = b; B.c = c;
    double M = fabs(a);
    if (fabs(b) > M) M = fabs(b);
    if (fabs(c) > M) M = fabs(c);
    const double U = 1.0 + M;   // Cauchy bound: every root has |s| < U
    const double f0 = c;        // f(0)
    const double disc = a * a - 3.0 * b;
    if (!(disc > 0.0)) {
        // monotone increasing: one real root, positive iff f(0) < 0
        if (f0 < 0.0) { B.lo = 0.0; B.hi = U; B.increasing = true; B.found = true; }
        return B;
    }
    const double sq = sqrt(disc);
    const double s1 = div3(-a - sq);   // local maximum: (-a - sq) / 3.0
    const double s2 = div3(-a + sq);   // local minimum: (-a + sq) / 3.0
    // (0, s1): increasing
    if (s1 > 0.0 && f0 < 0.0) {
        const double f1 = cubic_eval(a, b, c, s1);
        if (f1 >= 0.0) { B.lo = 0.0; B.hi = s1; B.increasing = true; B.found = true; return B; }
    }
    // (max(0,s1), s2): decreasing
    if (s2 > 0.0) {
        const double lo = s1 > 0.0 ? s1 : 0.0;
        const double fl = cubic_eval(a, b, c, lo);
        const double f2 = cubic_eval(a, b, c, s2);
        if (fl > 0.0 && f2 <= 0.0) { B.lo = lo; B.hi = s2; B.increasing = false; B.found = true; return B; }
    }
    // (max(0,s2), U): increasing
    {
        const double lo = s2 > 0.0 ? s2 : 0.0;
        const double fl = cubic_eval(a, b, c, lo);
        if (fl < 0.0) { B.lo = lo; B.hi = U; B.increasing = true; B.found = true; }
    }
    return B;
}

// Refinement by 64-way sectioning: every round evaluates the cubic at the 64
// interior points lo + (hi-lo)(l+1)/65 and keeps the sub-interval where the sign
// changes, until both ends round to the same float32.  Serial form (host, oracle,
// a lone device thread); k_post_step runs the same rounds with one lane per
// point (section_root_wave in cvo_kernels.hip) -- identical arithmetic per point,
// hence identical results.
constexpr int SECTIONS = 64;

CVO_HD double section_point(double lo, double w, int l)
{
    return lo + w * ((double)(l + 1) * (1.0 / 65.0));
}

// A short cut through the rounds, taken after the first one: two Newton steps from the middle of the
// bracket and a PROOF that the rounds would end on the same float32 -- or nothing (the rounds go on as if
// it had not been tried).  With u = 2^-53 the computed cubic_eval(s) is within 5u S(s) of the cubic,
// S(s) = ((s + |a|) s + |b|) s + |c|, for every 0 <= s <= hi; E = 16u S(hi).  If the computed values at
// xl = x - d and xr = x + d lie beyond -2E and +2E (in the bracket's direction) the cubic itself is beyond
// -E / +E there, and -- it is monotone on the bracket -- beyond them at every point left of xl / right of xr:
// every section point the rounds can still visit left of xl says "go right", every one right of xr says "stop".
// The rounds' final bracket therefore overlaps [xl, xr], and ends either with both ends on one float32
// (then that float32 lies between those of xl and xr) or with no section point left inside (then its
// upper end is within 65 ulps of a point of [xl, xr]): if xl - m and xr + m, m = 2^-44 |x|, round to the
// same float32, that is the float32 of the rounds' result.  The realistic cubics (traces of the oracle)
// take the short cut every time, after ONE round instead of 5.8 on average; random ones with wild
// coefficients fall back half the time (tests/test_host_math.py holds both against the plain rounds).
CVO_HD bool section_shortcut(const CubicBracket &B, double lo, double hi, double *root)
{
    double x = 0.5 * (lo + hi);
    for (int it = 0; it < 2; ++it) {
        const double d = (3.0 * x + 2.0 * B.a) * x + B.b;
#if defined(__HIP_DEVICE_COMPILE__)
        // (on the device the quotient is a reciprocal with one correction -- 2^-50 or so -- instead of the ~40 dependent instructions of
        // a float64 division, twice on the one chain of the head: the proof below holds for ANY x, so what comes out is the float32 of
        // the rounds' result whichever x went in; host and device may take the short cut with different x, or only one of them at all)
        double r = __builtin_amdgcn_rcp(d);
        r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
        x = x - cubic_eval(B.a, B.b, B.c, x) * r;
#else
        x = x - cubic_eval(B.a, B.b, B.c, x) / d;
#endif
    }
    if (!(fabs(x) <= 1.0e300)) return false;   // (NaN, infinity)
    const double delta = fabs(x) * 0x1p-40, m = fabs(x) * 0x1p-44;
    const double xl = x - delta, xr = x + delta;
    const double E = 0x1p-49 * (((hi + fabs(B.a)) * hi + fabs(B.b)) * hi + fabs(B.c));
    const double fl = cubic_eval(B.a, B.b, B.c, xl), fr = cubic_eval(B.a, B.b, B.c, xr);
    const bool crosses = B.increasing ? (fl < -2.0 * E && fr > 2.0 * E) : (fl > 2.0 * E && fr < -2.0 * E);
    if (!(crosses && xl > lo && xr < hi && (float)(xl - m) == (float)(xr + m))) return false;
    *root = xr;
    return true;
}

CVO_HD double section_root(const CubicBracket &B)
{
    double lo = B.lo, hi = B.hi;
    for (int round = 0; round < 64; ++round) {
        if ((float)lo == (float)hi) break;   // the root's float value is decided
        if (round == 1) {
            double r;
            if (section_shortcut(B, lo, hi, &r)) return r;
        }
        const double w = hi - lo;
        double nlo = lo, nhi = hi;
        for (int l = 0; l < SECTIONS; ++l) {
            const double x = section_point(lo, w, l);
            const bool inside = x > lo && x < hi;
            const double f = cubic_eval(B.a, B.b, B.c, x);
            const bool go_right = inside && (B.increasing ? (f < 0.0) : (f > 0.0));
            if (!go_right) {
                if (inside) nhi = x;
                break;
            }
            nlo = x;
        }
        if (nlo == lo && nhi == hi) break;   // no representable point left in between
        lo = nlo;
        hi = nhi;
    }
    return hi;
}

// min_step if there is no positive real root, clamp to 0.8 (ref cvo.cpp:298-307)
CVO_HD float finish_step(bool found, double root, float min_step)
{
    float step = min_step;
    if (found) {
        const float r = (float)root;
        if (r > 0.0f) step = r;
    }
    step = ((double)step > 0.8) ? (float)0.8 : step;
    return step;
}

CVO_HD float pick_step(const double bcde[4], float min_step)
{
    const CubicBracket B = cubic_bracket(bcde);
    return finish_step(B.found, B.found ? section_root(B) : 0.0, min_step);
}

// Exp_SEK3 with K = 1: dR (row-major) and dT = Jl * v.  In two stages (round 6): what depends on the twist alone -- the angle, the skew
// matrix and its square, the square roots of dist_se3 and of the stop test -- and what needs the step.  Inside a resident run the
// first stage is formed while the step-size sums are still on their way (cvo_kernels.hip kt_run); everywhere else the two stages
// follow each other: the same operations on the same operands either way.
struct ExpPre {
    float theta, theta2, theta3;
    Mat3 A, A2;
    double root_full, root_v;   // sqrt(2 |w|^2 + |v|^2), sqrt(|v|^2) in float64 (dist_se3)
    float nw, nv;               // float norms of the twist (the stop test of cvo, ref src/cvo.cpp:380)
    double nw_d, nv_d;          // ... in float64 of the float vectors (acvo, ref src/adaptive_cvo.cpp:509)
    int small;                  // theta < TOLERANCE: R = I, Jl = I
};

CVO_HD ExpPre exp_se3_pre(const float w[3], const float v[3])
{
    const float TOLERANCE = 1e-6f;
    ExpPre P;
    P.theta = norm_fixed3(w);
    P.small = (P.theta < TOLERANCE) ? 1 : 0;
    P.A = skew(w);
    P.theta2 = P.theta * P.theta;
    P.A2 = mul(P.A, P.A);
    P.theta3 = P.theta2 * P.theta;
    const double w2 = (double)w[0] * w[0] + (double)w[1] * w[1] + (double)w[2] * w[2];
    const double v2 = (double)v[0] * v[0] + (double)v[1] * v[1] + (double)v[2] * v[2];
    P.root_full = sqrt(2.0 * w2 + v2);
    P.root_v = sqrt(v2);
    P.nw = P.theta;
    P.nv = norm_fixed3(v);
    P.nw_d = sqrt((double)w[0] * w[0] + ((double)w[1] * w[1] + (double)w[2] * w[2]));
    P.nv_d = sqrt((double)v[0] * v[0] + ((double)v[1] * v[1] + (double)v[2] * v[2]));
    return P;
}

CVO_HD void exp_se3_with(const ExpPre &P, const float v[3], float dt, float dR[9], float dT[3])
{
    const Mat3 I = identity3();
    Mat3 R = I, Jl = I;   // small-angle branch: R = I, Jl = I (not dt*I)
    if (!P.small) {
        const float theta = P.theta;
        double sd, cd;
        sincos_det((double)(dt * theta), &sd, &cd);
        const float stheta = (float)sd;
        const float ctheta = (float)cd;
        const float oneMinusCosTheta2 = (1 - ctheta) / P.theta2;
        const float s1 = stheta / theta;
        const float j3 = (dt * theta - stheta) / P.theta3;
        for (int k = 0; k < 9; ++k) {
            R.m[k] = (I.m[k] + s1 * P.A.m[k]) + oneMinusCosTheta2 * P.A2.m[k];
            Jl.m[k] = (dt * I.m[k] + oneMinusCosTheta2 * P.A.m[k]) + j3 * P.A2.m[k];
        }
    }
    for (int k = 0; k < 9; ++k) dR[k] = R.m[k];
    mulv(Jl, v, dT);
}

CVO_HD void exp_se3(const float w[3], const float v[3], float dt, float dR[9], float dT[3])
{
    exp_se3_with(exp_se3_pre(w, v), v, dt, dR, dT);
}

// ||logm([dR dT; 0 1])||_F for the increment produced by exp_se3(w, v, dt).
CVO_HD float dist_se3_with(const ExpPre &P, float dt)
{
    if (P.theta < 1e-6f) return (float)P.root_v;
    return (float)((double)dt * P.root_full);
}

CVO_HD float dist_se3(const float w[3], const float v[3], float dt)
{
    return dist_se3_with(exp_se3_pre(w, v), dt);
}

// Constants of the per-point Taylor vectors in compute_step_size
// (ref src/cvo.cpp:226-238): omega_hat powers are left-associated products.
struct XiConsts {
    float omega[3], v[3];
    float W2[9], W3[9], W4[9];
    float u2[3], u3[3], u4[3];
};

CVO_HD XiConsts make_xi_consts(const float omega[3], const float v[3])
{
    XiConsts c;
    for (int k = 0; k < 3; ++k) { c.omega[k] = omega[k]; c.v[k] = v[k]; }
    const Mat3 W = skew(omega);
    const Mat3 W2 = mul(W, W);
    const Mat3 W3 = mul(W2, W);
    const Mat3 W4 = mul(W3, W);
    for (int k = 0; k < 9; ++k) { c.W2[k] = W2.m[k]; c.W3[k] = W3.m[k]; c.W4[k] = W4.m[k]; }
    mulv(W, v, c.u2);
    mulv(W2, v, c.u3);
    mulv(W3, v, c.u4);
    return c;
}

}   // namespace cvo_math
