// Which way does a real Eigen associate the three-term sums of its fixed-size float expressions?
//
// The oracle (oracle/orb_oracle.cpp, note at orc_is_in_frustum) and the kernels evaluate Vector3f::dot / norm and the
// coefficients of Matrix3f * Vector3f as e0 + (e1 + e2): Eigen's redux_novec_unroller (Eigen/src/Core/Redux.h) splits a range of
// Length terms at Length / 2.  Eigen is not in the build image, so that is restated from its published source.  On a machine with
// Eigen (any 3.3 / 3.4; the reference needs one):
//     g++ -O2 -ffp-contract=off -I/usr/include/eigen3 tests/tools/eigen_order_probe.cpp -o probe && ./probe
// prints, for dot(), squaredNorm() and a matrix-vector product, which of the two associations reproduces Eigen on inputs where they
// differ, and exits 0 iff all of them are e0 + (e1 + e2).  tests/test_eigen_order.py runs it when <Eigen/Dense> can be found and
// is skipped - saying so - otherwise.
#include <Eigen/Dense>
#include <cstdio>
#include <cstring>
#include <random>

static unsigned bits(float f) {
    unsigned u;
    std::memcpy(&u, &f, 4);
    return u;
}

int main() {
    std::mt19937 rng(7);
    std::uniform_real_distribution<float> U(-3.f, 3.f);
    int seen[3] = {0, 0, 0}, halves[3] = {0, 0, 0}, left[3] = {0, 0, 0};
    for (int it = 0; it < 200000; it++) {
        Eigen::Vector3f a(U(rng), U(rng), U(rng)), b(U(rng), U(rng), U(rng));
        Eigen::Matrix3f M;
        for (int i = 0; i < 9; i++) M(i / 3, i % 3) = U(rng);
        const volatile float p0 = a[0] * b[0], p1 = a[1] * b[1], p2 = a[2] * b[2];
        const float l = (p0 + p1) + p2, h = p0 + (p1 + p2);
        const float d = a.dot(b);
        if (bits(l) != bits(h)) { seen[0]++; halves[0] += bits(d) == bits(h); left[0] += bits(d) == bits(l); }
        const volatile float s0 = a[0] * a[0], s1 = a[1] * a[1], s2 = a[2] * a[2];
        const float ln = (s0 + s1) + s2, hn = s0 + (s1 + s2), n2 = a.squaredNorm();
        if (bits(ln) != bits(hn)) { seen[1]++; halves[1] += bits(n2) == bits(hn); left[1] += bits(n2) == bits(ln); }
        const Eigen::Vector3f y = M * b;
        const volatile float m0 = M(0, 0) * b[0], m1 = M(0, 1) * b[1], m2 = M(0, 2) * b[2];
        const float lm = (m0 + m1) + m2, hm = m0 + (m1 + m2);
        if (bits(lm) != bits(hm)) { seen[2]++; halves[2] += bits(y[0]) == bits(hm); left[2] += bits(y[0]) == bits(lm); }
    }
    const char *names[3] = {"Vector3f::dot", "Vector3f::squaredNorm", "(Matrix3f * Vector3f)[0]"};
    int ok = 1;
    for (int k = 0; k < 3; k++) {
        std::printf("%-26s cases where the associations differ: %d   e0 + (e1 + e2): %d   (e0 + e1) + e2: %d\n", names[k], seen[k], halves[k],
                    left[k]);
        ok &= seen[k] > 0 && halves[k] == seen[k];
    }
    std::printf("Eigen %d.%d.%d: %s\n", EIGEN_WORLD_VERSION, EIGEN_MAJOR_VERSION, EIGEN_MINOR_VERSION,
                ok ? "e0 + (e1 + e2) everywhere - as the oracle assumes" : "NOT the association the oracle assumes");
    return ok ? 0 : 1;
}
