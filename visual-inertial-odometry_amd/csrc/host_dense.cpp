// host_dense.cpp — see host_dense.h.  Plain C++17, no third-party linear algebra.
#include "host_dense.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

namespace vio_host {

// LU with partial pivoting, then solve against the identity: the route Eigen takes for a fixed 15x15
// `covariance.inverse()` (LU/PartialPivLU.h, LU/InverseImpl.h).
void inverse15(const double *cov, double *info) {
    constexpr int n = 15;
    double lu[n * n];
    int piv[n];
    std::memcpy(lu, cov, sizeof(lu));
    for (int k = 0; k < n; ++k) {
        int row = k;
        double big = std::fabs(lu[n * k + k]);
        for (int i = k + 1; i < n; ++i)
            if (std::fabs(lu[n * i + k]) > big) { big = std::fabs(lu[n * i + k]); row = i; }
        piv[k] = row;
        if (big != 0) {
            if (row != k)
                for (int j = 0; j < n; ++j) std::swap(lu[n * k + j], lu[n * row + j]);
            for (int i = k + 1; i < n; ++i) lu[n * i + k] /= lu[n * k + k];
        }
        for (int i = k + 1; i < n; ++i)
            for (int j = k + 1; j < n; ++j) lu[n * i + j] -= lu[n * i + k] * lu[n * k + j];
    }
    for (int c = 0; c < n; ++c) {
        double x[n];
        for (int i = 0; i < n; ++i) x[i] = (i == c) ? 1.0 : 0.0;
        for (int k = 0; k < n; ++k)
            if (piv[k] != k) std::swap(x[k], x[piv[k]]);
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < i; ++j) x[i] -= lu[n * i + j] * x[j];
        for (int i = n - 1; i >= 0; --i) {
            for (int j = i + 1; j < n; ++j) x[i] -= lu[n * i + j] * x[j];
            x[i] /= lu[n * i + i];
        }
        for (int i = 0; i < n; ++i) info[n * i + c] = x[i];
    }
}

// Householder tridiagonalisation followed by the implicit-shift QL iteration.
bool symmetric_eigen(int n, const double *Ain, double *d, double *Vout) {
    std::vector<double> Vv((size_t)n * n), ev(n);
    double *V = Vv.data(), *e = ev.data();
    auto at = [&](int i, int j) -> double & { return V[(size_t)i * n + j]; };
    for (int i = 0; i < n; ++i)
        for (int j = 0; j <= i; ++j) { at(i, j) = Ain[(size_t)i * n + j]; at(j, i) = at(i, j); }
    for (int j = 0; j < n; ++j) d[j] = at(n - 1, j);
    for (int i = n - 1; i > 0; --i) {
        double scale = 0.0, h = 0.0;
        for (int k = 0; k < i; ++k) scale += std::fabs(d[k]);
        if (scale == 0.0) {
            e[i] = d[i - 1];
            for (int j = 0; j < i; ++j) { d[j] = at(i - 1, j); at(i, j) = 0.0; at(j, i) = 0.0; }
        } else {
            for (int k = 0; k < i; ++k) { d[k] /= scale; h += d[k] * d[k]; }
            double f = d[i - 1];
            double g = std::sqrt(h);
            if (f > 0) g = -g;
            e[i] = scale * g;
            h -= f * g;
            d[i - 1] = f - g;
            for (int j = 0; j < i; ++j) e[j] = 0.0;
            for (int j = 0; j < i; ++j) {
                f = d[j];
                at(j, i) = f;
                g = e[j] + at(j, j) * f;
                for (int k = j + 1; k <= i - 1; ++k) { g += at(k, j) * d[k]; e[k] += at(k, j) * f; }
                e[j] = g;
            }
            f = 0.0;
            for (int j = 0; j < i; ++j) { e[j] /= h; f += e[j] * d[j]; }
            const double hh = f / (h + h);
            for (int j = 0; j < i; ++j) e[j] -= hh * d[j];
            for (int j = 0; j < i; ++j) {
                f = d[j]; g = e[j];
                for (int k = j; k <= i - 1; ++k) at(k, j) -= (f * e[k] + g * d[k]);
                d[j] = at(i - 1, j);
                at(i, j) = 0.0;
            }
        }
        d[i] = h;
    }
    for (int i = 0; i < n - 1; ++i) {
        at(n - 1, i) = at(i, i);
        at(i, i) = 1.0;
        const double h = d[i + 1];
        if (h != 0.0) {
            for (int k = 0; k <= i; ++k) d[k] = at(k, i + 1) / h;
            for (int j = 0; j <= i; ++j) {
                double g = 0.0;
                for (int k = 0; k <= i; ++k) g += at(k, i + 1) * at(k, j);
                for (int k = 0; k <= i; ++k) at(k, j) -= g * d[k];
            }
        }
        for (int k = 0; k <= i; ++k) at(k, i + 1) = 0.0;
    }
    for (int j = 0; j < n; ++j) { d[j] = at(n - 1, j); at(n - 1, j) = 0.0; }
    at(n - 1, n - 1) = 1.0;
    e[0] = 0.0;
    for (int i = 1; i < n; ++i) e[i - 1] = e[i];
    e[n - 1] = 0.0;
    double f = 0.0, tst1 = 0.0;
    const double eps = std::ldexp(1.0, -52);
    bool ok = true;
    for (int l = 0; l < n; ++l) {
        tst1 = std::max(tst1, std::fabs(d[l]) + std::fabs(e[l]));
        int m = l;
        while (m < n) { if (std::fabs(e[m]) <= eps * tst1) break; ++m; }
        if (m == n) m = n - 1;
        if (m > l) {
            int iter = 0;
            do {
                if (++iter > 200) { ok = false; break; }
                double g = d[l];
                double p = (d[l + 1] - g) / (2.0 * e[l]);
                double r = std::hypot(p, 1.0);
                if (p < 0) r = -r;
                d[l] = e[l] / (p + r);
                d[l + 1] = e[l] * (p + r);
                const double dl1 = d[l + 1];
                double h = g - d[l];
                for (int i = l + 2; i < n; ++i) d[i] -= h;
                f += h;
                p = d[m];
                double c = 1.0, c2 = c, c3 = c, s = 0.0, s2 = 0.0;
                const double el1 = e[l + 1];
                for (int i = m - 1; i >= l; --i) {
                    c3 = c2; c2 = c; s2 = s;
                    g = c * e[i];
                    h = c * p;
                    r = std::hypot(p, e[i]);
                    e[i + 1] = s * r;
                    s = e[i] / r;
                    c = p / r;
                    p = c * d[i] - s * g;
                    d[i + 1] = h + s * (c * g + s * d[i]);
                    for (int k = 0; k < n; ++k) {
                        h = at(k, i + 1);
                        at(k, i + 1) = s * at(k, i) + c * h;
                        at(k, i) = c * at(k, i) - s * h;
                    }
                }
                p = -s * s2 * c3 * el1 * e[l] / dl1;
                e[l] = s * p;
                d[l] = c * p;
            } while (std::fabs(e[l]) > eps * tst1);
        }
        d[l] += f;
        e[l] = 0.0;
    }
    for (int i = 0; i < n - 1; ++i) {
        int k = i;
        double p = d[i];
        for (int j = i + 1; j < n; ++j)
            if (d[j] < p) { k = j; p = d[j]; }
        if (k != i) {
            d[k] = d[i]; d[i] = p;
            for (int j = 0; j < n; ++j) std::swap(at(j, i), at(j, k));
        }
    }
    std::memcpy(Vout, V, sizeof(double) * (size_t)n * n);
    return ok;
}

static void move_to_bottom(std::vector<double> &H, std::vector<double> &b, int n, int idx, int dim) {
    std::vector<int> order;
    for (int i = 0; i < n; ++i)
        if (i < idx || i >= idx + dim) order.push_back(i);
    for (int i = idx; i < idx + dim; ++i) order.push_back(i);
    std::vector<double> T((size_t)n * n), tb(n);
    for (int i = 0; i < n; ++i) {
        for (int j = 0; j < n; ++j) T[(size_t)i * n + j] = H[(size_t)order[i] * n + order[j]];
        tb[i] = b[order[i]];
    }
    H.swap(T);
    b.swap(tb);
}

void marginalize_tail(double *Hin, double *bin, int frame, double *Hout, double *bout, double *errout, double *jtout) {
    const int n = 171, m2 = 15, n2 = n - m2;
    std::vector<double> H(Hin, Hin + (size_t)n * n), b(bin, bin + n);
    // larger index first: speed-bias, then pose (problem.cc:721-745)
    move_to_bottom(H, b, n, 12 + 15 * frame, 9);
    move_to_bottom(H, b, n, 6 + 15 * frame, 6);
    const double eps = 1e-8;
    double Amm[m2 * m2], ev[m2], V[m2 * m2], Ainv[m2 * m2];
    for (int i = 0; i < m2; ++i)
        for (int j = 0; j < m2; ++j) Amm[i * m2 + j] = 0.5 * (H[(size_t)(n2 + i) * n + n2 + j] + H[(size_t)(n2 + j) * n + n2 + i]);
    symmetric_eigen(m2, Amm, ev, V);
    for (int i = 0; i < m2; ++i)
        for (int j = 0; j < m2; ++j) {
            double s = 0;
            for (int k = 0; k < m2; ++k) s += V[i * m2 + k] * (ev[k] > eps ? 1.0 / ev[k] : 0.0) * V[j * m2 + k];
            Ainv[i * m2 + j] = s;
        }
    std::vector<double> tempB((size_t)n2 * m2), Hp((size_t)n2 * n2), bp(n2);
    for (int i = 0; i < n2; ++i)
        for (int j = 0; j < m2; ++j) {
            double s = 0;
            for (int k = 0; k < m2; ++k) s += H[(size_t)i * n + n2 + k] * Ainv[k * m2 + j];
            tempB[(size_t)i * m2 + j] = s;
        }
    for (int i = 0; i < n2; ++i) {
        for (int j = 0; j < n2; ++j) {
            double s = 0;
            for (int k = 0; k < m2; ++k) s += tempB[(size_t)i * m2 + k] * H[(size_t)(n2 + k) * n + j];
            Hp[(size_t)i * n2 + j] = H[(size_t)i * n + j] - s;
        }
        double s = 0;
        for (int k = 0; k < m2; ++k) s += tempB[(size_t)i * m2 + k] * b[n2 + k];
        bp[i] = b[i] - s;
    }
    std::vector<double> ev2(n2), V2((size_t)n2 * n2);
    symmetric_eigen(n2, Hp.data(), ev2.data(), V2.data());
    for (int i = 0; i < n2; ++i) {
        const double sinv = ev2[i] > eps ? std::sqrt(1.0 / ev2[i]) : 0.0;
        for (int j = 0; j < n2; ++j) jtout[(size_t)i * n2 + j] = sinv * V2[(size_t)j * n2 + i];
    }
    for (int i = 0; i < n2; ++i) {
        double s = 0;
        for (int j = 0; j < n2; ++j) s += -jtout[(size_t)i * n2 + j] * bp[j];
        errout[i] = s;
    }
    for (int i = 0; i < n2; ++i)
        for (int j = 0; j < n2; ++j) {
            double s = 0;
            for (int k = 0; k < n2; ++k) {
                const double sk = ev2[k] > eps ? ev2[k] : 0.0;
                s += V2[(size_t)i * n2 + k] * sk * V2[(size_t)j * n2 + k];
            }
            Hout[(size_t)i * n2 + j] = std::fabs(s) > 1e-9 ? s : 0.0;     // problem.cc:778
        }
    std::memcpy(bout, bp.data(), sizeof(double) * n2);
}

}  // namespace vio_host
