// se3_host.hpp -- the O(1) host mathematics of one align() iteration.
//
// Everything here is per-iteration scalar work: the inverse transform
// (ref src/cvo.cpp:83-87), kernel thresholds (ref src/cvo.cpp:102-103), the
// step-size cubic (ref src/cvo.cpp:53-69,291-307), Exp_SEK3
// (ref src/LieGroup.cpp:159-186), dist_se3 (ref src/cvo.cpp:71-81) and the
// length-scale schedules (ref src/cvo.cpp:408-410, src/adaptive_cvo.cpp:538-545).
//
// Arithmetic contract: float32 expressions are evaluated in
// Eigen's coefficient order with NO contraction (this translation unit is
// compiled with -ffp-contract=off); transcendental functions are evaluated in
// float64 and rounded once to float32.
#pragma once

#include <cmath>
#include <cstring>
#include <limits>

namespace cvo_host {

struct Mat3 {
    float m[9];   // row-major
    float &operator()(int r, int c) { return m[3 * r + c]; }
    float operator()(int r, int c) const { return m[3 * r + c]; }
};

inline Mat3 identity3()
{
    Mat3 I{};
    I.m[0] = I.m[4] = I.m[8] = 1.0f;
    return I;
}

inline Mat3 mul(const Mat3 &a, const Mat3 &b)
{
    Mat3 o{};
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c)
            o(r, c) = (a(r, 0) * b(0, c) + a(r, 1) * b(1, c)) + a(r, 2) * b(2, c);
    return o;
}

inline void mul(const Mat3 &a, const float x[3], float out[3])
{
    float t[3];
    for (int r = 0; r < 3; ++r) t[r] = (a(r, 0) * x[0] + a(r, 1) * x[1]) + a(r, 2) * x[2];
    std::memcpy(out, t, sizeof(t));
}

inline Mat3 skew(const float w[3])
{ // ref src/LieGroup.cpp:20-27
    Mat3 M{};
    M(0, 1) = -w[2]; M(0, 2) = w[1];
    M(1, 0) = w[2];  M(1, 2) = -w[0];
    M(2, 0) = -w[1]; M(2, 1) = w[0];
    return M;
}

// Vector3f::squaredNorm() (fixed size 3, non-vectorised unrolled redux)
inline float sqnorm_fixed3(const float a[3]) { return a[0] * a[0] + (a[1] * a[1] + a[2] * a[2]); }
inline float norm_fixed3(const float a[3]) { return std::sqrt(sqnorm_fixed3(a)); }

// [Rt | t] = [R^T | -R^T T]
inline void inverse_tf(const float R[9], const float T[3], float Rt[9], float t[3])
{
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) Rt[3 * r + c] = R[3 * c + r];
    for (int r = 0; r < 3; ++r)
        t[r] = ((-Rt[3 * r]) * T[0] + (-Rt[3 * r + 1]) * T[1]) + (-Rt[3 * r + 2]) * T[2];
}

inline void tf_to_mat4(const float Rt[9], const float t[3], float m[16])
{
    for (int k = 0; k < 16; ++k) m[k] = 0.0f;
    for (int r = 0; r < 3; ++r) {
        for (int c = 0; c < 3; ++c) m[4 * r + c] = Rt[3 * r + c];
        m[4 * r + 3] = t[r];
    }
    m[15] = 1.0f;
}

inline void mat4_mul(const float a[16], const float b[16], float out[16])
{
    float t[16];
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c)
            t[4 * r + c] = ((a[4 * r] * b[c] + a[4 * r + 1] * b[4 + c]) + a[4 * r + 2] * b[8 + c]) +
                           a[4 * r + 3] * b[12 + c];
    std::memcpy(out, t, sizeof(t));
}

inline float log_f32(float x) { return (float)std::log((double)x); }
inline float sin_f32(float x) { return (float)std::sin((double)x); }
inline float cos_f32(float x) { return (float)std::cos((double)x); }

// d2_thres / d2_c_thres: `float = -2.0*l*l*log(sp/s2)` with a float log.
inline float d2_threshold(float l, float sp, float s2)
{
    return (float)(-2.0 * l * l * (double)log_f32(sp / s2));
}
inline float d2c_threshold(float c_ell, float c_sp, float c_sigma)
{
    return (float)(-2.0 * c_ell * c_ell * (double)log_f32(c_sp / c_sigma / c_sigma));
}

// Smallest positive real root of 4E s^3 + 3D s^2 + 2C s + B, else min_step;
// clamp to 0.8.  Coefficients are rounded to float exactly as the reference
// stores them in its VectorXf; the roots are found in closed form in float64
// (the reference runs a float companion-matrix eigen-solve) and rounded.
inline float pick_step(const double bcde[4], float min_step)
{
    const float c3 = (float)(4.0 * (float)bcde[3]);
    const float c2 = (float)(3.0 * (float)bcde[2]);
    const float c1 = (float)(2.0 * (float)bcde[1]);
    const float c0 = (float)bcde[0];
    float best = std::numeric_limits<float>::infinity();
    if (c3 != 0.0f && std::isfinite(c3) && std::isfinite(c2) && std::isfinite(c1) &&
        std::isfinite(c0)) {
        const double a = (double)(c2 / c3), b = (double)(c1 / c3), c = (double)(c0 / c3);
        const double Q = (a * a - 3.0 * b) / 9.0;
        const double Rr = (2.0 * a * a * a - 9.0 * a * b + 27.0 * c) / 54.0;
        double roots[3];
        int nr = 0;
        if (Rr * Rr < Q * Q * Q) {
            const double sq = std::sqrt(Q);
            double ct = Rr / (sq * sq * sq);
            ct = ct > 1.0 ? 1.0 : (ct < -1.0 ? -1.0 : ct);
            const double th = std::acos(ct);
            const double two_pi = 6.283185307179586476925286766559;
            roots[0] = -2.0 * sq * std::cos(th / 3.0) - a / 3.0;
            roots[1] = -2.0 * sq * std::cos((th + two_pi) / 3.0) - a / 3.0;
            roots[2] = -2.0 * sq * std::cos((th - two_pi) / 3.0) - a / 3.0;
            nr = 3;
        } else {
            const double s = std::sqrt(Rr * Rr - Q * Q * Q);
            double A = -std::cbrt(std::fabs(Rr) + s);
            if (Rr < 0) A = -A;
            const double Bq = (A != 0.0) ? Q / A : 0.0;
            roots[0] = (A + Bq) - a / 3.0;
            nr = 1;
        }
        for (int i = 0; i < nr; ++i) {
            double s = roots[i];
            for (int it = 0; it < 2; ++it) {   // Newton polish on the monic cubic
                const double f = ((s + a) * s + b) * s + c;
                const double fp = (3.0 * s + 2.0 * a) * s + b;
                if (fp != 0.0 && std::isfinite(f / fp)) s -= f / fp;
            }
            const float r = (float)s;
            if (r > 0 && r < best) best = r;
        }
    }
    float step = (best == std::numeric_limits<float>::infinity()) ? min_step : best;
    step = step > 0.8 ? (float)0.8 : step;
    return step;
}

// Exp_SEK3 with K = 1: returns dR (row-major) and dT = Jl * v.
inline void exp_se3(const float w[3], const float v[3], float dt, float dR[9], float dT[3])
{
    const float TOLERANCE = 1e-6f;
    const float theta = norm_fixed3(w);
    const Mat3 I = identity3();
    Mat3 R = I, Jl = I;   // small-angle branch: R = I, Jl = I (not dt*I)
    if (!(theta < TOLERANCE)) {
        const Mat3 A = skew(w);
        const float theta2 = theta * theta;
        const float stheta = sin_f32(dt * theta);
        const float ctheta = cos_f32(dt * theta);
        const float oneMinusCosTheta2 = (1 - ctheta) / theta2;
        const Mat3 A2 = mul(A, A);
        const float s1 = stheta / theta;
        const float j3 = (dt * theta - stheta) / (theta2 * theta);
        for (int k = 0; k < 9; ++k) {
            R.m[k] = (I.m[k] + s1 * A.m[k]) + oneMinusCosTheta2 * A2.m[k];
            Jl.m[k] = (dt * I.m[k] + oneMinusCosTheta2 * A.m[k]) + j3 * A2.m[k];
        }
    }
    std::memcpy(dR, R.m, sizeof(R.m));
    mul(Jl, v, dT);
}

// ||logm([dR dT; 0 1])||_F for the increment produced by exp_se3(w, v, dt).
inline float dist_se3(const float w[3], const float v[3], float dt)
{
    const double w2 = (double)w[0] * w[0] + (double)w[1] * w[1] + (double)w[2] * w[2];
    const double v2 = (double)v[0] * v[0] + (double)v[1] * v[1] + (double)v[2] * v[2];
    if (norm_fixed3(w) < 1e-6f) return (float)std::sqrt(v2);
    return (float)((double)dt * std::sqrt(2.0 * w2 + v2));
}

// Constants of the per-point Taylor vectors in compute_step_size
// (ref src/cvo.cpp:226-238): omega_hat powers are left-associated products.
struct XiConsts {
    float omega[3], v[3];
    float W2[9], W3[9], W4[9];
    float u2[3], u3[3], u4[3];
};

inline XiConsts make_xi_consts(const float omega[3], const float v[3])
{
    XiConsts c{};
    std::memcpy(c.omega, omega, sizeof(c.omega));
    std::memcpy(c.v, v, sizeof(c.v));
    const Mat3 W = skew(omega);
    const Mat3 W2 = mul(W, W);
    const Mat3 W3 = mul(W2, W);
    const Mat3 W4 = mul(W3, W);
    std::memcpy(c.W2, W2.m, sizeof(c.W2));
    std::memcpy(c.W3, W3.m, sizeof(c.W3));
    std::memcpy(c.W4, W4.m, sizeof(c.W4));
    mul(W, v, c.u2);
    mul(W2, v, c.u3);
    mul(W3, v, c.u4);
    return c;
}

}   // namespace cvo_host
