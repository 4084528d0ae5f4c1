// Host-side precompute: quadrature rules, Legendre tables, FFT twiddles.
// float64 arithmetic, rounded once to fp32 -- mirrors what the reference gets from
// torch-harmonics' numpy precompute followed by `.float()` (sfnonet.py:536-539).
#include "common.h"
#include "../../include/makani_amd.h"

#include <cmath>
#include <cstring>
#include <vector>

namespace mk {
static thread_local std::string g_err;
void set_error(const std::string& msg) { g_err = msg; }
}  // namespace mk

extern "C" int mk_version(void) { return 100; }
extern "C" const char* mk_last_error(void) { return mk::g_err.c_str(); }

namespace {

// Clenshaw-Curtis on cos(theta): nodes cos(linspace(pi, 0, n)), ascending in x.
void clenshaw_curtis(int n, std::vector<double>& x, std::vector<double>& w) {
    x.resize(n);
    w.resize(n);
    const int n1 = n - 1;
    const double step = (0.0 - M_PI) / n1;
    for (int j = 0; j < n; ++j) x[j] = std::cos(j == n1 ? 0.0 : M_PI + j * step);
    if (n == 2) {
        w[0] = w[1] = 1.0;
        return;
    }
    for (int j = 0; j < n; ++j) {
        const double th = M_PI * j / n1;
        double s = 1.0;
        for (int k = 1; k <= n1 / 2; ++k) {
            const double bk = (2 * k == n1) ? 1.0 : 2.0;
            s -= bk / (4.0 * k * k - 1.0) * std::cos(2.0 * k * th);
        }
        const double c = (j == 0 || j == n1) ? 1.0 : 2.0;
        w[j] = c / n1 * s;
    }
}

// Gauss-Legendre by Newton iteration on P_n; ascending nodes.
void gauss_legendre(int n, std::vector<double>& x, std::vector<double>& w) {
    x.resize(n);
    w.resize(n);
    for (int i = 0; i < (n + 1) / 2; ++i) {
        double z = std::cos(M_PI * (i + 0.75) / (n + 0.5));
        double pp = 0.0;
        for (int it = 0; it < 100; ++it) {
            double p1 = 1.0, p2 = 0.0;
            for (int j = 1; j <= n; ++j) {
                const double p3 = p2;
                p2 = p1;
                p1 = ((2.0 * j - 1.0) * z * p2 - (j - 1.0) * p3) / j;
            }
            pp = n * (z * p1 - p2) / (z * z - 1.0);
            const double dz = p1 / pp;
            z -= dz;
            if (std::fabs(dz) < 1e-16) break;
        }
        // recompute derivative at the converged node
        double p1 = 1.0, p2 = 0.0;
        for (int j = 1; j <= n; ++j) {
            const double p3 = p2;
            p2 = p1;
            p1 = ((2.0 * j - 1.0) * z * p2 - (j - 1.0) * p3) / j;
        }
        pp = n * (z * p1 - p2) / (z * z - 1.0);
        x[i] = -z;
        x[n - 1 - i] = z;
        w[i] = w[n - 1 - i] = 2.0 / ((1.0 - z * z) * pp * pp);
    }
}

int nodes_weights(int grid, int nlat, std::vector<double>& theta, std::vector<double>& w) {
    std::vector<double> x;
    if (grid == 0)
        clenshaw_curtis(nlat, x, w);
    else if (grid == 1)
        gauss_legendre(nlat, x, w);
    else
        return 1;
    // colatitudes: flip(arccos(x)) -> ascending from the north pole; weights are symmetric
    theta.resize(nlat);
    for (int k = 0; k < nlat; ++k) theta[k] = std::acos(x[nlat - 1 - k]);
    return 0;
}

}  // namespace

extern "C" int mk_quadrature(int grid, int nlat, double* theta, double* weights) {
    MK_REQUIRE(nlat >= 2, "nlat must be >= 2");
    std::vector<double> t, w;
    MK_REQUIRE(nodes_weights(grid, nlat, t, w) == 0, "unknown grid (0 = equiangular, 1 = legendre-gauss)");
    std::memcpy(theta, t.data(), sizeof(double) * nlat);
    std::memcpy(weights, w.data(), sizeof(double) * nlat);
    return 0;
}

extern "C" int mk_legendre_kpad(int nlat) { return (nlat + 31) / 32 * 32; }

extern "C" int mk_legendre_table(int grid, int nlat, int lmax, int mmax, int with_quad_weights, float* out) {
    MK_REQUIRE(nlat >= 2 && lmax >= 1 && mmax >= 1, "bad sizes");
    std::vector<double> theta, w;
    MK_REQUIRE(nodes_weights(grid, nlat, theta, w) == 0, "unknown grid (0 = equiangular, 1 = legendre-gauss)");
    const int K = nlat, KP = mk_legendre_kpad(nlat);
    const int nmax = lmax > mmax ? lmax : mmax;
    std::vector<double> cost(K), diag(K), prev2(K), prev1(K), cur(K);
    for (int k = 0; k < K; ++k) cost[k] = std::cos(theta[k]);
    std::memset(out, 0, sizeof(float) * (size_t)mmax * lmax * KP);
    // diag = P[m][m]; advanced by the diagonal recursion as m grows
    for (int k = 0; k < K; ++k) diag[k] = 1.0 / std::sqrt(4.0 * M_PI);
    for (int m = 0; m < mmax && m < nmax; ++m) {
        if (m > 0) {
            const int l = m;
            for (int k = 0; k < K; ++k)
                diag[k] = std::sqrt((2.0 * l + 1.0) * (1.0 + cost[k]) * (1.0 - cost[k]) / 2.0 / l) * diag[k];
        }
        const double sgn = (m & 1) ? -1.0 : 1.0;  // Condon-Shortley phase
        float* om = out + (size_t)m * lmax * KP;
        auto emit = [&](int l, const std::vector<double>& v) {
            if (l >= lmax) return;
            float* o = om + (size_t)l * KP;
            for (int k = 0; k < K; ++k) o[k] = (float)(sgn * v[k] * (with_quad_weights ? w[k] : 1.0));
        };
        prev2 = diag;  // P[m][m]
        emit(m, prev2);
        if (m + 1 < nmax) {
            const int l = m + 1;
            for (int k = 0; k < K; ++k) prev1[k] = std::sqrt(2.0 * l + 1.0) * cost[k] * prev2[k];
            emit(l, prev1);
        }
        for (int l = m + 2; l < lmax; ++l) {
            const double a = std::sqrt((2.0 * l - 1.0) / (l - m) * (2.0 * l + 1.0) / (l + m));
            const double b = std::sqrt((double)(l + m - 1) / (l - m) * (2.0 * l + 1.0) / (2.0 * l - 3.0) * (l - m - 1) / (l + m));
            for (int k = 0; k < K; ++k) cur[k] = cost[k] * a * prev1[k] - b * prev2[k];
            emit(l, cur);
            prev2.swap(prev1);
            prev1.swap(cur);
        }
    }
    return 0;
}

extern "C" int mk_fft_twiddle_len(int nlon) { return 2 * (nlon / 2) + 2 * (nlon / 2 + 1); }

extern "C" int mk_fft_twiddles(int nlon, float* out) {
    MK_REQUIRE(nlon >= 2 && nlon % 2 == 0, "nlon must be even");
    const int H = nlon / 2;
    for (int j = 0; j < H; ++j) {
        const double a = -2.0 * M_PI * j / H;
        out[2 * j] = (float)std::cos(a);
        out[2 * j + 1] = (float)std::sin(a);
    }
    float* u = out + 2 * H;
    for (int m = 0; m <= H; ++m) {
        const double a = -2.0 * M_PI * m / nlon;
        u[2 * m] = (float)std::cos(a);
        u[2 * m + 1] = (float)std::sin(a);
    }
    return 0;
}
