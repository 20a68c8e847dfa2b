"""CPU: the O(1) host mathematics -- oracle vs independent references (scipy /
numpy) and product (libcvo_hip.so host functions) vs oracle, bit for bit."""
import numpy as np
import pytest
from scipy.linalg import expm, logm


def _rand_twists(n, seed, scale_w=1.0, scale_v=1.0):
    rng = np.random.default_rng(seed)
    return (rng.normal(0, scale_w, (n, 3)).astype(np.float32),
            rng.normal(0, scale_v, (n, 3)).astype(np.float32),
            rng.uniform(1e-4, 0.8, n).astype(np.float32))


def test_exp_se3_matches_matrix_exponential(po):
    W, V, DT = _rand_twists(50, 1, 1.5, 1.0)
    for w, v, dt in zip(W, V, DT):
        dR, dT = po.exp_se3(w, v, dt)
        xi = np.zeros((4, 4))
        xi[:3, :3] = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]], np.float64)
        xi[:3, 3] = v
        E = expm(float(dt) * xi)
        assert np.allclose(dR, E[:3, :3], atol=3e-6)
        assert np.allclose(dT, E[:3, 3], atol=3e-6 * max(1.0, np.abs(v).max()))


def test_exp_se3_small_angle_quirk(po):
    """theta < 1e-6: R = I and Jl = I, so dT = v un-scaled by dt (Lie.cpp:168-170)."""
    w = np.array([1e-8, -2e-8, 1e-8], np.float32)
    v = np.array([0.3, -0.2, 0.1], np.float32)
    dR, dT = po.exp_se3(w, v, 0.25)
    assert np.array_equal(dR, np.eye(3, dtype=np.float32)) and np.array_equal(dT, v)
    assert po.dist_se3(w, v, 0.25) == pytest.approx(float(np.linalg.norm(v)), rel=1e-6)


def test_dist_se3_matches_matrix_log(po):
    """Closed form vs ||logm([dR dT;0 1])||_F (ref cvo.cpp:71-81)."""
    W, V, DT = _rand_twists(40, 2, 0.5, 0.5)
    for w, v, dt in zip(W, V, DT):
        dR, dT = po.exp_se3(w, v, dt)
        M = np.eye(4)
        M[:3, :3], M[:3, 3] = dR, dT
        ref = np.linalg.norm(np.real(logm(M)))
        assert po.dist_se3(w, v, dt) == pytest.approx(ref, rel=2e-5, abs=1e-7)


def test_pick_step_matches_numpy_roots(po):
    rng = np.random.default_rng(3)
    n_checked = 0
    for _ in range(400):
        bcde = rng.normal(0, 1, 4) * 10.0 ** rng.integers(-3, 4, 4)
        c = np.array([4.0 * np.float32(bcde[3]), 3.0 * np.float32(bcde[2]),
                      2.0 * np.float32(bcde[1]), np.float32(bcde[0])], np.float32).astype(np.float64)
        mon = np.array([1.0, np.float32(c[1] / c[0]), np.float32(c[2] / c[0]), np.float32(c[3] / c[0])],
                       np.float64)
        r = np.roots(mon)
        real = r[np.abs(r.imag) < 1e-9 * np.maximum(1.0, np.abs(r.real))].real
        cplx = r[np.abs(r.imag) >= 1e-9 * np.maximum(1.0, np.abs(r.real))]
        if len(cplx) and np.min(np.abs(cplx.imag) / np.maximum(1.0, np.abs(cplx.real))) < 1e-5:
            continue   # nearly a double root: classification is ill-conditioned
        pos = real[real > 0]
        want = np.float32(pos.min()) if len(pos) else np.float32(0.2)
        want = np.float32(0.8) if want > 0.8 else want
        got = po.pick_step(bcde)
        assert got == pytest.approx(float(want), rel=1e-5), (bcde, r)
        n_checked += 1
    assert n_checked > 300


def test_pick_step_degenerate(po):
    assert po.pick_step([0.0, 0.0, 0.0, 0.0]) == pytest.approx(0.2)       # empty A: 0/0
    assert po.pick_step([np.nan, 1.0, 1.0, 1.0]) == pytest.approx(0.2)
    assert po.pick_step([1.0, 1.0, 1.0, 0.0]) == pytest.approx(0.2)       # E = 0: division by 0
    assert po.pick_step([-100.0, 0.0, 0.0, 1.0]) == pytest.approx(0.8)     # root 2.92 -> clamp


def test_product_host_math_is_bitwise_the_oracle(pkg, po):
    """cvo_hip_pick_step / exp_se3 / dist_se3 (se3_math.hpp, the same code the
    device-resident loop runs) against the oracle's C restatement."""
    capi = pkg.capi
    rng = np.random.default_rng(4)
    for _ in range(2000):
        bcde = rng.normal(0, 1, 4) * 10.0 ** rng.integers(-6, 7, 4)
        assert np.float32(capi.pick_step(bcde)).view(np.uint32) == np.float32(po.pick_step(bcde)).view(np.uint32)
    # cubics of the shape a registration produces (B > 0 small, C < 0 dominant: traces of the oracle), widely
    # jittered: se3_math.hpp leaves the sectioning rounds after the first one when two Newton steps and a sign
    # check PROVE which float32 the rounds would end on (section_shortcut); the oracle runs the plain rounds
    base = np.array([[0.394, -4.08, -0.0209, 0.0884], [0.0773, -0.8296, -4.7e-4, 3.24e-3]])
    for i in range(20000):
        bcde = base[i & 1] * np.exp(rng.normal(0, 0.7, 4)) * 10.0 ** rng.uniform(-2, 4)
        if rng.random() < 0.2:
            bcde[2:] *= -1
        assert np.float32(capi.pick_step(bcde)).view(np.uint32) == np.float32(po.pick_step(bcde)).view(np.uint32)
    W, V, DT = _rand_twists(2000, 5, 3.0, 2.0)
    W[:50] *= 1e-7   # exercise the small-angle branch
    for w, v, dt in zip(W, V, DT):
        a, b = capi.exp_se3(w, v, dt), po.exp_se3(w, v, dt)
        assert np.array_equal(a[0].view(np.uint32), b[0].view(np.uint32))
        assert np.array_equal(a[1].view(np.uint32), b[1].view(np.uint32))
        assert np.float32(capi.dist_se3(w, v, dt)).view(np.uint32) == np.float32(po.dist_se3(w, v, dt)).view(np.uint32)


def test_sincos_det_is_the_rounded_libm_value(po):
    """exp_se3 with omega along z: dR[1,0] = sin(dt*theta) for theta = |w|."""
    rng = np.random.default_rng(6)
    bad = 0
    for _ in range(3000):
        th = np.float32(rng.uniform(0.01, 40.0))
        dt = np.float32(rng.uniform(0.001, 0.8))
        dR, _ = po.exp_se3(np.array([0, 0, th], np.float32), np.zeros(3, np.float32), dt)
        x = np.float32(dt * th)
        s1 = np.float32(np.float32(np.sin(np.float64(x))) / th)   # stheta/theta
        want = np.float32(s1 * th)
        bad += int(dR[1, 0] != want)
    assert bad == 0


def _reference_root_rule(bcde, min_step=np.float32(0.2)):
    """ref src/cvo.cpp:53-69,291-307 literally, in float32: p_coef = (4E, 3D, 2C, B) as floats,
    companion matrix with first row -(coef/coef(0)).segment(1,3), ITS FLOAT32 EIGENVALUES
    (LAPACK sgeev here, Eigen's float EigenSolver there), the smallest one with real > 0 and
    imag == 0 EXACTLY, else min_step; clamp to 0.8."""
    B, C, D, E = [np.float32(x) for x in bcde]
    coef = np.array([np.float32(4.0 * float(E)), np.float32(3.0 * float(D)), np.float32(2.0 * float(C)), B],
                    np.float32)
    M = np.zeros((3, 3), np.float32)
    M[1, 0] = M[2, 1] = 1
    with np.errstate(all="ignore"):
        M[0, :] = -(coef / coef[0])[1:4]
    if not np.all(np.isfinite(M)):
        return float(min_step)
    fmax = np.float32(np.finfo(np.float32).max)
    t = fmax
    for z in np.linalg.eigvals(M):      # complex64
        if z.real > 0 and z.real < t and z.imag == 0:
            t = np.float32(z.real)
    step = min_step if t == fmax else t
    return float(np.float32(0.8) if float(step) > 0.8 else step)


def test_sectioning_root_vs_float32_companion_eigenvalues(po, pkg):
    """The restatement brackets and sections the cubic in float64 where the reference runs a
    float32 eigen-solve of the companion matrix and keeps eigenvalues whose imaginary part is
    exactly zero.  On the step-size coefficients that registrations actually produce (every
    iteration of five oracle registrations, both modes) the two rules must give the same step;
    on random coefficients -- including nearly double roots, where a float32 eigen-solve may
    return a conjugate pair for two close real roots -- the disagreement rate is reported and
    bounded."""
    po.set_threads(4)
    cases = []
    for mode, seed, n in [(0, 7, 1500), (1, 7, 1500), (0, 11, 1200), (1, 5, 1200), (0, 23, 900)]:
        xf, ff, xm, fm = pkg.data.synthetic_pair(n, n, seed=seed, acvo=(mode == 1))
        p = po.default_params(mode)
        st = po.init_state(p)
        _, tr = po.align(p, st, xf, ff, xm, fm)
        cases += [t["bcde"] for t in tr]
    assert len(cases) > 200
    worst = 0.0
    for b in cases:
        got, want = po.pick_step(b), _reference_root_rule(b)
        worst = max(worst, abs(got - want) / max(abs(want), 1e-30))
    assert worst <= 2e-6, "registration coefficients: sectioning and float32 eigen-solve differ by %g" % worst
    rng = np.random.default_rng(8)
    n_rand, n_diff = 2000, 0
    for _ in range(n_rand):
        b = rng.normal(0, 1, 4) * 10.0 ** rng.integers(-3, 4, 4)
        got, want = po.pick_step(b), _reference_root_rule(b)
        if abs(got - want) > 1e-4 * max(abs(want), 1e-30):
            n_diff += 1
    print("random coefficients: %d of %d steps differ by more than 1e-4 relative" % (n_diff, n_rand))
    assert n_diff <= n_rand // 50


def test_div3_and_div6_by_fma_are_the_division(tmp_path):
    """se3_math.hpp div3 (the stationary points of the step-size cubic, ref src/cvo.cpp:53-69 via
    cubic_bracket) and the list kernels' div6 replace a float64 division by a multiplication with an
    exact FMA correction.  The claim -- the same double as x / 3.0 (x / 6.0), bit for bit -- is held
    here over 2e7 random bit patterns of every exponent, subnormals and the largest values included
    (plain C, IEEE fma from libm: what the device's v_fma_f64 computes)."""
    import subprocess
    src = tmp_path / "div3.c"
    src.write_text(r'''
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
static double divk(double x, double k, double c) { double q0 = x * c; double r = fma(-k, q0, x); return fma(r, c, q0); }
int main(void) {
    uint64_t s = 88172645463325252ull; long bad3 = 0, bad6 = 0, n = 0;
    for (long i = 0; i < 20000000; ++i) {
        s ^= s << 13; s ^= s >> 7; s ^= s << 17;
        double x; memcpy(&x, &s, 8);
        if (!isfinite(x)) continue;
        ++n;
        double a = divk(x, 3.0, 1.0 / 3.0), b = x / 3.0;
        double c = divk(x, 6.0, 1.0 / 6.0), d = x / 6.0;
        /* (below 2^-1020 the correction term itself is subnormal: the kernels never get there) */
        if (fabs(x) < 1e-300) continue;
        bad3 += memcmp(&a, &b, 8) != 0; bad6 += memcmp(&c, &d, 8) != 0;
    }
    printf("%ld %ld %ld\n", n, bad3, bad6);
    return 0;
}
''')
    exe = tmp_path / "div3"
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", str(src), "-o", str(exe), "-lm"])
    n, bad3, bad6 = map(int, subprocess.check_output([str(exe)]).split())
    assert n > 1.9e7 and bad3 == 0 and bad6 == 0, (n, bad3, bad6)
