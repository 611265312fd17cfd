/*
 * pols_oracle.c -- CPU ORACLE (test infrastructure, NOT product code).
 * See pols_oracle.h for scope, pinning status and the usage rule.
 *
 * Reference: azmyrajab/polars_ols v0.4.1.  "ls.rs" below = src/least_squares.rs,
 * "ex.rs" = src/expressions.rs, "ls.py" = polars_ols/least_squares.py,
 * "st.rs" = src/statistics.rs.
 *
 * Third-party arithmetic the reference delegates to (crate sources are not in
 * the reference tree) is restated from the published algorithms:
 *   faer 0.18.2  col_piv_qr / cholesky / partial_piv_lu / thin_svd
 *                -> column-pivoted Householder QR, LL^T, LU with partial
 *                   pivoting, one-sided Jacobi SVD (any backward-stable SVD
 *                   yields the same min-norm solution to O(cond*eps)).
 *   LAPACK dgelsd (via ndarray-linalg 0.16 / intel-mkl-src 0.8.1)
 *                -> minimum-norm least squares with rcond = machine epsilon.
 *   statrs 0.17.1 StudentsT::cdf -> regularised incomplete beta function.
 */
#include "pols_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------ utils */

static double *dalloc(size_t n) { return (double *)malloc(sizeof(double) * (n ? n : 1)); }

/* Grow-only per-thread workspaces for the per-group hot loop of the batched driver: the reference allocates per
 * call too, but through jemalloc (src/lib.rs:174-176); 100+ OpenMP threads hammering glibc malloc would make the
 * CPU baseline look worse than the reference really is. */
#define ORC_TLS_SLOTS 8
static __thread double *tls_buf[ORC_TLS_SLOTS];
static __thread size_t tls_cap[ORC_TLS_SLOTS];
static double *tls_alloc(int slot, size_t n) {
    if (n > tls_cap[slot]) {
        free(tls_buf[slot]);
        tls_cap[slot] = n + n / 4 + 64;
        tls_buf[slot] = (double *)malloc(sizeof(double) * tls_cap[slot]);
    }
    return tls_buf[slot];
}

int orc_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* faer `cholesky(Side::Lower)` (ls.rs:23,289): plain LL^T, Err on a
 * non-positive (or NaN) pivot.  On an exactly singular matrix the last pivot
 * is 0 in exact arithmetic and +/- a few ulps of a_jj as computed -- a coin
 * flip between "failed" and a division by rounding noise.  The reference
 * holds one vector for that case: the collinear frame of the demo notebook,
 * cell 30, prints "Cholesky decomposition failed, falling back to LU" and
 * {null, null, null}.  A pivot within 16 k eps of a_jj therefore counts as
 * non-positive here (and in the HIP kernels), which pins that outcome instead
 * of leaving it to the summation order. */
static int chol_factor(double *l, int k) {
    const double tol = 16.0 * (double)k * 2.220446049250313e-16;
    for (int j = 0; j < k; ++j) {
        double d = l[j * k + j];
        const double ajj = d;
        for (int p = 0; p < j; ++p) d -= l[j * k + p] * l[j * k + p];
        if (!(d > tol * ajj)) return 1;
        d = sqrt(d);
        l[j * k + j] = d;
        for (int i = j + 1; i < k; ++i) {
            double s = l[i * k + j];
            for (int p = 0; p < j; ++p) s -= l[i * k + p] * l[j * k + p];
            l[i * k + j] = s / d;
        }
    }
    return 0;
}

static void chol_solve_inplace(const double *l, int k, double *b) {
    for (int i = 0; i < k; ++i) {
        double s = b[i];
        for (int p = 0; p < i; ++p) s -= l[i * k + p] * b[p];
        b[i] = s / l[i * k + i];
    }
    for (int i = k - 1; i >= 0; --i) {
        double s = b[i];
        for (int p = i + 1; p < k; ++p) s -= l[p * k + i] * b[p];
        b[i] = s / l[i * k + i];
    }
}

int orc_cholesky_solve(const double *a, int k, const double *b, double *x) {
    double *l = dalloc((size_t)k * k);
    memcpy(l, a, sizeof(double) * k * k);
    int rc = chol_factor(l, k);
    if (rc == 0) {
        memcpy(x, b, sizeof(double) * k);
        chol_solve_inplace(l, k, x);
    }
    free(l);
    return rc;
}

/* faer `partial_piv_lu()` (ls.rs:34,267): LU with row partial pivoting.
 * Solves for nrhs right-hand sides stored as columns of b (k x nrhs row-major). */
static int lu_solve_multi(const double *a, int k, const double *b, int nrhs, double *x) {
    double *lu = dalloc((size_t)k * k);
    int *piv = (int *)malloc(sizeof(int) * (k ? k : 1));
    memcpy(lu, a, sizeof(double) * k * k);
    memcpy(x, b, sizeof(double) * k * nrhs);
    for (int j = 0; j < k; ++j) {
        int p = j;
        double best = fabs(lu[j * k + j]);
        for (int i = j + 1; i < k; ++i)
            if (fabs(lu[i * k + j]) > best) { best = fabs(lu[i * k + j]); p = i; }
        piv[j] = p;
        if (p != j) {
            for (int c = 0; c < k; ++c) { double t = lu[j * k + c]; lu[j * k + c] = lu[p * k + c]; lu[p * k + c] = t; }
            for (int c = 0; c < nrhs; ++c) { double t = x[j * nrhs + c]; x[j * nrhs + c] = x[p * nrhs + c]; x[p * nrhs + c] = t; }
        }
        double d = lu[j * k + j];
        for (int i = j + 1; i < k; ++i) {
            double f = lu[i * k + j] / d;
            lu[i * k + j] = f;
            for (int c = j + 1; c < k; ++c) lu[i * k + c] -= f * lu[j * k + c];
            for (int c = 0; c < nrhs; ++c) x[i * nrhs + c] -= f * x[j * nrhs + c];
        }
    }
    for (int i = k - 1; i >= 0; --i) {
        for (int c = 0; c < nrhs; ++c) {
            double s = x[i * nrhs + c];
            for (int p = i + 1; p < k; ++p) s -= lu[i * k + p] * x[p * nrhs + c];
            x[i * nrhs + c] = s / lu[i * k + i];
        }
    }
    free(lu);
    free(piv);
    return 0;
}

/* ls.rs:264-273 solve_ols_lu */
int orc_lu_solve(const double *a, int k, const double *b, double *x) {
    return lu_solve_multi(a, k, b, 1, x);
}

/* ls.rs:20-39 inv: Cholesky inverse, falling back to LU inverse. */
int orc_inv(const double *a, int k, int use_cholesky, double *out) {
    double *eye = dalloc((size_t)k * k);
    memset(eye, 0, sizeof(double) * k * k);
    for (int i = 0; i < k; ++i) eye[i * k + i] = 1.0;
    if (use_cholesky) {
        double *l = dalloc((size_t)k * k);
        memcpy(l, a, sizeof(double) * k * k);
        if (chol_factor(l, k) == 0) {
            double *col = dalloc(k);
            for (int c = 0; c < k; ++c) {
                for (int i = 0; i < k; ++i) col[i] = eye[i * k + c];
                chol_solve_inplace(l, k, col);
                for (int i = 0; i < k; ++i) out[i * k + c] = col[i];
            }
            free(col); free(l); free(eye);
            return 0;
        }
        free(l);
    }
    lu_solve_multi(a, k, eye, k, out);
    free(eye);
    return 0;
}

/* ls.rs:600-607 */
void orc_outer_product(const double *u, const double *v, int k, double *out) {
    for (int i = 0; i < k; ++i)
        for (int j = 0; j < k; ++j) out[i * k + j] = u[i] * v[j];
}

/* C = A(m x p) * B(p x q), all row-major */
static void matmul(const double *a, const double *b, int m, int p, int q, double *c) {
    for (int i = 0; i < m; ++i)
        for (int j = 0; j < q; ++j) {
            double s = 0.0;
            for (int t = 0; t < p; ++t) s += a[i * p + t] * b[t * q + j];
            c[i * q + j] = s;
        }
}

/* ls.rs:629-648 woodbury_update: (A + U C V)^-1 = A^-1 - A^-1 U (C^-1 + V A^-1 U)^-1 V A^-1
 * a_inv k x k, u k x r, c r x r, v r x k. */
void orc_woodbury_update(const double *a_inv, const double *u, const double *c, const double *v,
                         int k, int r, int c_is_diag, double *out) {
    double *inv_c = dalloc((size_t)r * r);
    if (c_is_diag) { /* inv_diag, ls.rs:609-617 */
        memset(inv_c, 0, sizeof(double) * r * r);
        for (int i = 0; i < r; ++i) inv_c[i * r + i] = 1.0 / c[i * r + i];
    } else {
        orc_inv(c, r, 0, inv_c);
    }
    double *v_inv_a = dalloc((size_t)r * k), *inv_a_u = dalloc((size_t)k * r);
    double *mid = dalloc((size_t)r * r), *mid_inv = dalloc((size_t)r * r);
    double *t1 = dalloc((size_t)k * r), *t2 = dalloc((size_t)k * k);
    matmul(v, a_inv, r, k, k, v_inv_a);
    matmul(a_inv, u, k, k, r, inv_a_u);
    matmul(v, inv_a_u, r, k, r, mid);
    for (int i = 0; i < r * r; ++i) mid[i] += inv_c[i];
    orc_inv(mid, r, 0, mid_inv);
    matmul(inv_a_u, mid_inv, k, r, r, t1);
    matmul(t1, v_inv_a, k, r, k, t2);
    for (int i = 0; i < k * k; ++i) out[i] = a_inv[i] - t2[i];
    free(inv_c); free(v_inv_a); free(inv_a_u); free(mid); free(mid_inv); free(t1); free(t2);
}

/* ls.rs:651-666 update_xtx_inv: x_update is r x k; U = x_update^T, V = x_update; C default I. */
void orc_update_xtx_inv(const double *xtx_inv, const double *x_update, const double *c_or_null,
                        int k, int r, double *out) {
    double *u = dalloc((size_t)k * r), *eye = dalloc((size_t)r * r);
    for (int i = 0; i < r; ++i)
        for (int j = 0; j < k; ++j) u[j * r + i] = x_update[i * k + j];
    memset(eye, 0, sizeof(double) * r * r);
    for (int i = 0; i < r; ++i) eye[i * r + i] = 1.0;
    orc_woodbury_update(xtx_inv, u, c_or_null ? c_or_null : eye, x_update, k, r, 1, out);
    free(u); free(eye);
}

/* ------------------------------------------------------------ QR solver */

/* ls.rs:195-205 solve_ols_qr: faer col_piv_qr().solve_lstsq -- Householder QR
 * with column pivoting (largest remaining column norm, the FIRST of tied
 * columns), then R z = (Q^T y)[:k], beta = P z.  Requires n >= k.
 *
 * Rank-deficient X: the reference holds one printed vector for it -- the
 * collinear frame of notebooks/polars_ols_demo.ipynb cell 28 (x3 an exact copy
 * of x2, y = x1 + x2 + x3) prints {1.0, 2.0, -0.0} for solve_method="qr": the
 * BASIC solution, the dependent column at exactly zero (cell 32: only "svd"
 * gives the minimum-norm {1, 1, 1}).  A textbook back substitution through the
 * rounding-noise pivot R_jj ~ eps |R_00| instead returns noise / noise
 * ({1, 0.976, 1.024} on that frame).  This restatement is therefore
 * rank-revealing the way LAPACK dgelsy is: the factorisation stops at the first
 * pivot with |R_jj| <= eps * max(n, k) * |R_00| (the cut-off the reference
 * itself uses for singular values, ls.rs:143-145), the columns from there on
 * get coefficient 0 and the leading block is back-substituted.  Pinned by
 * tests/golden/notebook_kat.json (cells 28 / 32). */
void orc_solve_ols_qr(const double *y, const double *x, int64_t n, int k, double *beta) {
    double *a = tls_alloc(0, (size_t)n * k); /* column-major */
    double *b = tls_alloc(1, (size_t)n);
    int jpvt_small[64];
    int *jpvt = (k <= 64) ? jpvt_small : (int *)malloc(sizeof(int) * k);
    for (int64_t i = 0; i < n; ++i) {
        b[i] = y[i];
        for (int j = 0; j < k; ++j) a[(size_t)j * n + i] = x[i * k + j];
    }
    for (int j = 0; j < k; ++j) jpvt[j] = j;
    int steps = (int)((n < k) ? n : k);
    int rank = steps;
    double r00 = 0.0;
    const double rank_tol = 2.220446049250313e-16 * (double)((n > k) ? n : k);
    for (int j = 0; j < steps; ++j) {
        /* pivot selection on the trailing sub-columns */
        int best = j;
        double bestn = -1.0;
        for (int c = j; c < k; ++c) {
            double s = 0.0;
            const double *col = a + (size_t)c * n;
            for (int64_t i = j; i < n; ++i) s += col[i] * col[i];
            if (s > bestn) { bestn = s; best = c; }
        }
        if (best != j) {
            double *c1 = a + (size_t)j * n, *c2 = a + (size_t)best * n;
            for (int64_t i = 0; i < n; ++i) { double t = c1[i]; c1[i] = c2[i]; c2[i] = t; }
            int t = jpvt[j]; jpvt[j] = jpvt[best]; jpvt[best] = t;
        }
        double *cj = a + (size_t)j * n;
        double normx = sqrt(bestn);
        if (j == 0) r00 = normx;
        if (normx <= rank_tol * r00) { rank = j; break; } /* numerically dependent from here on (NaN data: never true) */
        double alpha = cj[j];
        double bh = -copysign(normx, alpha);
        double tau = (bh - alpha) / bh;
        double scale = 1.0 / (alpha - bh);
        for (int64_t i = j + 1; i < n; ++i) cj[i] *= scale;
        cj[j] = bh;
        /* apply H = I - tau v v^T (v_j = 1) to the remaining columns and to b */
        for (int c = j + 1; c <= k; ++c) {
            double *col = (c < k) ? a + (size_t)c * n : b;
            double w = col[j];
            for (int64_t i = j + 1; i < n; ++i) w += cj[i] * col[i];
            w *= tau;
            col[j] -= w;
            for (int64_t i = j + 1; i < n; ++i) col[i] -= w * cj[i];
        }
    }
    double *z = tls_alloc(2, (size_t)k);
    for (int i = k - 1; i >= rank; --i) z[i] = 0.0;
    for (int i = rank - 1; i >= 0; --i) {
        double s = b[i];
        for (int p = i + 1; p < rank; ++p) s -= a[(size_t)p * n + i] * z[p];
        z[i] = s / a[(size_t)i * n + i];
    }
    for (int i = 0; i < k; ++i) beta[jpvt[i]] = z[i];
    if (jpvt != jpvt_small) free(jpvt);
}

/* ------------------------------------------------------------ SVD solvers */

/* One-sided Jacobi SVD of a tall matrix W (rows x cols, rows >= cols), column
 * major.  On exit the columns of W are U*diag(s); V (cols x cols, column-major)
 * accumulates the rotations. */
static void jacobi_svd_tall(double *w, int64_t rows, int cols, double *v, double *s) {
    for (int i = 0; i < cols * cols; ++i) v[i] = 0.0;
    for (int i = 0; i < cols; ++i) v[i * cols + i] = 1.0;
    for (int sweep = 0; sweep < 60; ++sweep) {
        int rotated = 0;
        for (int p = 0; p < cols - 1; ++p) {
            for (int q = p + 1; q < cols; ++q) {
                double *wp = w + (size_t)p * rows, *wq = w + (size_t)q * rows;
                double a = 0.0, b = 0.0, g = 0.0;
                for (int64_t i = 0; i < rows; ++i) { a += wp[i] * wp[i]; b += wq[i] * wq[i]; g += wp[i] * wq[i]; }
                if (g == 0.0 || fabs(g) <= 1e-15 * sqrt(a * b)) continue;
                rotated = 1;
                double zeta = (b - a) / (2.0 * g);
                double t = copysign(1.0, zeta) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                double c = 1.0 / sqrt(1.0 + t * t), sn = c * t;
                for (int64_t i = 0; i < rows; ++i) {
                    double xp = wp[i], xq = wq[i];
                    wp[i] = c * xp - sn * xq;
                    wq[i] = sn * xp + c * xq;
                }
                double *vp = v + (size_t)p * cols, *vq = v + (size_t)q * cols;
                for (int i = 0; i < cols; ++i) {
                    double xp = vp[i], xq = vq[i];
                    vp[i] = c * xp - sn * xq;
                    vq[i] = sn * xp + c * xq;
                }
            }
        }
        if (!rotated) break;
    }
    for (int j = 0; j < cols; ++j) {
        double nn = 0.0;
        const double *wj = w + (size_t)j * rows;
        for (int64_t i = 0; i < rows; ++i) nn += wj[i] * wj[i];
        s[j] = sqrt(nn);
    }
}

/* beta(k x m) = V * diag(d(s)) * U^T * y(n x m) for x (n x k), any shape.
 * mode 0: d = 1/s for s > cutoff else 0            (min-norm LS, dgelsd)
 * mode 1: d = s/(s^2+alpha) after zeroing s<cutoff (ls.rs:143-148) */
static void svd_apply(const double *y, const double *x, int64_t n, int k, int m, int mode,
                      double alpha, double rc_factor, double *beta) {
    int tall = (n >= k);
    int64_t rows = tall ? n : k;
    int cols = tall ? k : (int)n;
    double *w = dalloc((size_t)rows * cols), *v = dalloc((size_t)cols * cols), *s = dalloc(cols);
    if (tall) {
        for (int64_t i = 0; i < n; ++i)
            for (int j = 0; j < k; ++j) w[(size_t)j * rows + i] = x[i * k + j];
    } else { /* SVD of x^T (k x n) */
        for (int64_t i = 0; i < n; ++i)
            for (int j = 0; j < k; ++j) w[(size_t)i * rows + j] = x[i * k + j];
    }
    jacobi_svd_tall(w, rows, cols, v, s);
    double smax = 0.0;
    for (int j = 0; j < cols; ++j) if (s[j] > smax) smax = s[j];
    double cutoff = rc_factor * smax;
    for (int64_t i = 0; i < (int64_t)k * m; ++i) beta[i] = 0.0;
    double *coefs = dalloc(m);
    for (int j = 0; j < cols; ++j) {
        double sj = s[j], d;
        if (mode == 0) {
            d = (sj > cutoff && sj > 0.0) ? 1.0 / sj : 0.0;
        } else {
            double sz = (sj < cutoff) ? 0.0 : sj;
            d = sz / (sz * sz + alpha);
        }
        if (d == 0.0) continue;
        const double *wj = w + (size_t)j * rows; /* = u_j * s_j */
        const double *vj = v + (size_t)j * cols;
        if (tall) {
            /* x = U S V^T: left vectors in w (n), right vectors in v (k) */
            for (int t = 0; t < m; ++t) {
                double acc = 0.0;
                for (int64_t i = 0; i < n; ++i) acc += wj[i] * y[i * m + t];
                coefs[t] = (sj > 0.0) ? d * acc / sj : 0.0;
            }
            for (int i = 0; i < k; ++i)
                for (int t = 0; t < m; ++t) beta[i * m + t] += vj[i] * coefs[t];
        } else {
            /* x^T = U' S V'^T  =>  x = V' S U'^T: left vectors of x are v (n), right are w/s (k) */
            for (int t = 0; t < m; ++t) {
                double acc = 0.0;
                for (int64_t i = 0; i < n; ++i) acc += vj[i] * y[i * m + t];
                coefs[t] = (sj > 0.0) ? d * acc / sj : 0.0;
            }
            for (int i = 0; i < k; ++i)
                for (int t = 0; t < m; ++t) beta[i * m + t] += wj[i] * coefs[t];
        }
    }
    free(w); free(v); free(s); free(coefs);
}

/* ls.rs:183-191 solve_ols_svd on linux-x86_64: LAPACK dgelsd via
 * ndarray-linalg `least_squares` (rcond argument ignored, :181; LAPACK default
 * rcond = machine epsilon).  Minimum-norm solution.
 * (An EXACTLY dependent column is a knife edge at this cut-off in any SVD: its
 * singular value comes out as a few eps * s_max, growing with the column count --
 * {1, 1, 1} on the 3-column frame of the demo notebook, cell 32, rounding noise
 * divided by rounding noise at 40 columns, in LAPACK as here.  Raising the cut-off
 * to numpy's eps * max(n, k) would settle that but break a case the reference's
 * tests DO hold: test_fit_multi_collinear[99-"svd"], n = k = 100 with a 1e-12
 * shift, resolves a direction at ~5e-14 of s_max.) */
void orc_solve_ols_svd(const double *y, const double *x, int64_t n, int k, int m, double *beta) {
    svd_apply(y, x, n, k, m, 0, 0.0, DBL_EPSILON, beta);
}

/* ls.rs:106-168 solve_ridge_svd */
void orc_solve_ridge_svd(const double *y, const double *x, int64_t n, int k, int m, double alpha,
                         int has_rcond, double rcond, double *beta) {
    double rc = has_rcond ? rcond : DBL_EPSILON * (double)((n > k) ? n : k); /* :143-144 */
    svd_apply(y, x, n, k, m, 1, alpha, rc, beta);
}

/* ls.rs:211-240 solve_ols */
int orc_solve_ols(const double *y, const double *x, int64_t n, int k, int method, double *beta) {
    int use_qr;
    if (method == ORC_METHOD_QR) use_qr = 1;
    else if (method == ORC_METHOD_SVD) use_qr = 0;
    else if (method == ORC_METHOD_NONE) use_qr = (n > k); /* :225-229 */
    else return -1;                                       /* panic :231 */
    if (use_qr) orc_solve_ols_qr(y, x, n, k, beta);
    else orc_solve_ols_svd(y, x, n, k, 1, beta);
    return 0;
}

/* ls.rs:243-260 solve_multi_target; y is n x m, beta is k x m */
void orc_solve_multi_target(const double *y, const double *x, int64_t n, int k, int m, double alpha,
                            int has_rcond, double rcond, double *beta) {
    if (n == 0 || k == 0) { /* :250-252 */
        for (int64_t i = 0; i < (int64_t)k * m; ++i) beta[i] = 0.0;
        return;
    }
    if (alpha > 0.0) orc_solve_ridge_svd(y, x, n, k, m, alpha, has_rcond, rcond, beta);
    else orc_solve_ols_svd(y, x, n, k, m, beta);
}

/* ls.rs:277-337 solve_normal_equations (xtx k x k, xty k). */
int orc_solve_normal_equations(const double *xtx, const double *xty, int k, int method,
                               int fallback, double *beta) {
    if (method == ORC_METHOD_NONE) method = ORC_METHOD_CHOL; /* :284 */
    switch (method) {
    case ORC_METHOD_CHOL:
        if (orc_cholesky_solve(xtx, k, xty, beta) == 0) return 0;
        if (fallback == ORC_METHOD_NONE) fallback = ORC_METHOD_SVD; /* :301 */
        if (fallback == ORC_METHOD_SVD) { orc_solve_ols_svd(xty, xtx, k, k, 1, beta); return 1; }
        if (fallback == ORC_METHOD_LU) { orc_lu_solve(xtx, k, xty, beta); return 1; }
        if (fallback == ORC_METHOD_QR) { orc_solve_ols_qr(xty, xtx, k, k, beta); return 1; }
        return -2; /* panic :322 */
    case ORC_METHOD_LU:
        orc_lu_solve(xtx, k, xty, beta);
        return 0;
    case ORC_METHOD_QR:
    case ORC_METHOD_SVD:
        return orc_solve_ols(xty, xtx, k, k, method, beta); /* :334 */
    default:
        return -3; /* panic :335 */
    }
}

/* ls.rs:342-371 solve_ridge */
int orc_solve_ridge(const double *y, const double *x, int64_t n, int k, double alpha, int method,
                    int has_rcond, double rcond, double *beta) {
    if (!(alpha >= 0.0)) return -4; /* assert :349 */
    if (method == ORC_METHOD_CHOL || method == ORC_METHOD_LU || method == ORC_METHOD_NONE) {
        double *xtx = dalloc((size_t)k * k), *xty = dalloc(k);
        for (int a = 0; a < k; ++a) {
            for (int b = 0; b < k; ++b) {
                double s = 0.0;
                for (int64_t i = 0; i < n; ++i) s += x[i * k + a] * x[i * k + b];
                xtx[a * k + b] = s;
            }
            double s = 0.0;
            for (int64_t i = 0; i < n; ++i) s += x[i * k + a] * y[i];
            xty[a] = s;
        }
        for (int a = 0; a < k; ++a) xtx[a * k + a] += alpha; /* :355-356 */
        int rc = orc_solve_normal_equations(xtx, xty, k, method, ORC_METHOD_LU, beta); /* :358-363 */
        free(xtx); free(xty);
        return rc < 0 ? rc : 0;
    }
    if (method == ORC_METHOD_SVD) {
        orc_solve_ridge_svd(y, x, n, k, 1, alpha, has_rcond, rcond, beta);
        return 0;
    }
    return -5; /* panic :366 */
}

/* ls.rs:373-379 */
static double soft_threshold(double x, double alpha, int positive) {
    double mag = fabs(x) - alpha;
    if (mag < 0.0) mag = 0.0;
    double sgn = (x > 0.0) ? 1.0 : ((x < 0.0) ? -1.0 : (signbit(x) ? -1.0 : 1.0)); /* f64::signum */
    double r = sgn * mag;
    if (positive && r < 0.0) r = 0.0;
    return r;
}

/* ls.rs:386-492 solve_elastic_net: residual-form cyclic coordinate descent. */
int orc_solve_elastic_net(const double *y, const double *x, int64_t n, int k, double alpha,
                          int has_l1_ratio, double l1_ratio, int64_t max_iter, double tol,
                          int positive, int method, double *w, int64_t *n_iter_out) {
    if (!has_l1_ratio) l1_ratio = 0.5;                  /* :396 */
    if (method == ORC_METHOD_NONE) method = ORC_METHOD_CD; /* :400 */
    if (method != ORC_METHOD_CD && method != ORC_METHOD_CD_ACTIVE_SET) return -6; /* :404 */
    if (!(alpha > 0.0)) return -7;                      /* :409 */
    if (!(l1_ratio >= 0.0 && l1_ratio <= 1.0)) return -8; /* :410-413 */

    double *xc = dalloc((size_t)n * k); /* column-major copy: x.slice(s![.., j]) */
    double *diag = dalloc(k), *res = dalloc((size_t)n), *w_old = dalloc(k);
    for (int j = 0; j < k; ++j) {
        double s = 0.0;
        for (int64_t i = 0; i < n; ++i) { double v = x[i * k + j]; xc[(size_t)j * n + i] = v; s += v * v; }
        diag[j] = s; /* xtx[[j,j]], :417,:431 */
        w[j] = 0.0;
    }
    for (int64_t i = 0; i < n; ++i) res[i] = y[i];
    alpha = alpha * (double)n; /* :419 */
    const double thr = alpha * l1_ratio, l2 = alpha * (1.0 - l1_ratio);
    int *active = (int *)malloc(sizeof(int) * (k ? k : 1)), *iter_list = (int *)malloc(sizeof(int) * (k ? k : 1));
    int n_active = k;
    for (int j = 0; j < k; ++j) active[j] = j;
    int64_t it = 0;
    for (; it < max_iter; ++it) {
        memcpy(w_old, w, sizeof(double) * k);
        int n_iter_list = (method == ORC_METHOD_CD) ? k : n_active;
        for (int t = 0; t < n_iter_list; ++t) iter_list[t] = (method == ORC_METHOD_CD) ? t : active[t]; /* clone :459 */
        for (int t = 0; t < n_iter_list; ++t) {
            int j = iter_list[t];
            const double *xj = xc + (size_t)j * n;
            double wj = w[j], dot = 0.0;
            for (int64_t i = 0; i < n; ++i) res[i] = res[i] + xj[i] * wj;       /* :428 */
            for (int64_t i = 0; i < n; ++i) dot += xj[i] * res[i];               /* :430 */
            wj = soft_threshold(dot, thr, positive) / (diag[j] + l2);           /* :430-431 */
            w[j] = wj;
            for (int64_t i = 0; i < n; ++i) res[i] = res[i] - xj[i] * wj;       /* :433 */
            if (method == ORC_METHOD_CD_ACTIVE_SET && fabs(wj) < tol) {          /* :472-476 */
                for (int q = 0; q < n_active; ++q)
                    if (active[q] == j) {
                        memmove(active + q, active + q + 1, sizeof(int) * (n_active - q - 1));
                        --n_active;
                        break;
                    }
            }
        }
        double d2 = 0.0;
        for (int j = 0; j < k; ++j) d2 += (w[j] - w_old[j]) * (w[j] - w_old[j]);
        if (sqrt(d2) < tol) { ++it; break; } /* :436-444 */
    }
    if (n_iter_out) *n_iter_out = it;
    free(xc); free(diag); free(res); free(w_old); free(active); free(iter_list);
    return 0;
}

/* ex.rs:351-388 _get_least_squares_coefficients */
int orc_get_coefficients(const double *y, const double *x, int64_t n, int k,
                         const orc_ols_params *p, double *beta) {
    if (n == 0 || k == 0) { /* features.is_empty(), :357-359 */
        for (int j = 0; j < k; ++j) beta[j] = 0.0;
        return 0;
    }
    const int m = p->solve_method;
    if (p->alpha == 0.0 && !p->positive &&
        (m == ORC_METHOD_NONE || m == ORC_METHOD_SVD || m == ORC_METHOD_QR))
        return orc_solve_ols(y, x, n, k, m, beta);
    if (p->alpha >= 0.0 && (p->has_l1_ratio ? p->l1_ratio : 0.0) == 0.0 && !p->positive)
        return orc_solve_ridge(y, x, n, k, p->alpha, m, p->has_rcond, p->rcond, beta);
    return orc_solve_elastic_net(y, x, n, k, p->alpha, p->has_l1_ratio, p->l1_ratio, p->max_iter,
                                 p->tol, p->positive, m, beta, NULL);
}

/* -------------------------------------------------------------------- RLS */

/* ls.rs:494-598 */
void orc_solve_rls(const double *y, const double *x, int64_t n, int k, int has_half_life,
                   double half_life, double initial_state_covariance,
                   const double *initial_state_mean_or_null, const uint8_t *is_valid,
                   double *coef_out) {
    const double ff = has_half_life ? exp(log(0.5) / half_life) : 1.0; /* :513-517 */
    double *coef = dalloc(k), *P = dalloc((size_t)k * k), *kg = dalloc(k), *xtp = dalloc(k), *px = dalloc(k);
    for (int i = 0; i < k; ++i) coef[i] = initial_state_mean_or_null ? initial_state_mean_or_null[i] : 0.0;
    for (int i = 0; i < k * k; ++i) P[i] = 0.0;
    for (int i = 0; i < k; ++i) P[i * k + i] = initial_state_covariance; /* :520 */
    for (int64_t t = 0; t < n; ++t) {
        const double *xt = x + t * k;
        if (!is_valid || is_valid[t]) { /* update, :531-540 */
            for (int j = 0; j < k; ++j) { double s = 0.0; for (int i = 0; i < k; ++i) s += xt[i] * P[i * k + j]; xtp[j] = s; }
            double q = 0.0;
            for (int j = 0; j < k; ++j) q += xtp[j] * xt[j];
            const double r = 1.0 + q / ff;
            for (int i = 0; i < k; ++i) { double s = 0.0; for (int j = 0; j < k; ++j) s += P[i * k + j] * xt[j]; px[i] = s; }
            const double den = r * ff;
            for (int i = 0; i < k; ++i) kg[i] = px[i] / den;
            double pr = 0.0;
            for (int i = 0; i < k; ++i) pr += xt[i] * coef[i];
            const double resid = y[t] - pr;
            for (int i = 0; i < k; ++i) coef[i] = coef[i] + kg[i] * resid;
            for (int i = 0; i < k; ++i)
                for (int j = 0; j < k; ++j) P[i * k + j] = P[i * k + j] / ff - (kg[i] * kg[j]) * r;
        }
        for (int i = 0; i < k; ++i) coef_out[t * k + i] = coef[i]; /* :592-594 */
    }
    free(coef); free(P); free(kg); free(xtp); free(px);
}

/* ---------------------------------------------------------------- rolling */

typedef struct {
    int k, woodbury;
    double *m;   /* xtx (non-woodbury) or xtx_inv (woodbury), k x k */
    double *xty; /* k */
} roll_state;

static void roll_rank1(roll_state *s, const double *x, double y, double sign) {
    const int k = s->k;
    for (int i = 0; i < k; ++i) {
        for (int j = 0; j < k; ++j) s->m[i * k + j] += sign * (x[i] * x[j]);
        s->xty[i] = s->xty[i] + sign * (x[i] * y);
    }
}

/* NonWoodburyState::update ls.rs:707-725 / WoodburyState::update :749-776 */
static void roll_update(roll_state *s, const double *xn, double yn, const double *xp, double yp, int has_prev) {
    const int k = s->k;
    if (!s->woodbury) {
        roll_rank1(s, xn, yn, +1.0);
        if (has_prev) roll_rank1(s, xp, yp, -1.0);
        return;
    }
    double *out = dalloc((size_t)k * k);
    if (has_prev) {
        double *xu = dalloc((size_t)2 * k);
        for (int j = 0; j < k; ++j) { xu[j] = -xp[j]; xu[k + j] = xn[j]; } /* :762-765 */
        const double c[4] = {-1.0, 0.0, 0.0, 1.0};                        /* :744 */
        orc_update_xtx_inv(s->m, xu, c, k, 2, out);
        for (int j = 0; j < k; ++j) s->xty[j] = s->xty[j] + xn[j] * yn - xp[j] * yp;
        free(xu);
    } else {
        orc_update_xtx_inv(s->m, xn, NULL, k, 1, out);
        for (int j = 0; j < k; ++j) s->xty[j] = s->xty[j] + xn[j] * yn;
    }
    memcpy(s->m, out, sizeof(double) * k * k);
    free(out);
}

/* ::subtract ls.rs:727-730 / :778-782 */
static void roll_subtract(roll_state *s, const double *xp, double yp) {
    const int k = s->k;
    if (!s->woodbury) { roll_rank1(s, xp, yp, -1.0); return; }
    for (int j = 0; j < k; ++j) s->xty[j] = s->xty[j] - xp[j] * yp;
    double *out = dalloc((size_t)k * k);
    const double c[1] = {-1.0};
    orc_update_xtx_inv(s->m, xp, c, k, 1, out);
    memcpy(s->m, out, sizeof(double) * k * k);
    free(out);
}

/* ::solve ls.rs:732-734 (Cholesky -> LU) / :784-786 */
static void roll_solve(roll_state *s, double *beta) {
    const int k = s->k;
    if (!s->woodbury) { orc_solve_normal_equations(s->m, s->xty, k, ORC_METHOD_NONE, ORC_METHOD_LU, beta); return; }
    for (int i = 0; i < k; ++i) { double a = 0.0; for (int j = 0; j < k; ++j) a += s->m[i * k + j] * s->xty[j]; beta[i] = a; }
}

/* ls.rs:848-1032 solve_rolling_ols */
void orc_solve_rolling_ols(const double *y, const double *x, int64_t n, int k, int64_t window_size,
                           int64_t min_periods_in, int use_woodbury_in, double alpha,
                           const uint8_t *is_valid_in, int null_policy, double *coef_out) {
    const int64_t min_periods = (min_periods_in >= 0) ? min_periods_in : ((k < window_size) ? k : window_size); /* :860 */
    const int use_woodbury = (use_woodbury_in >= 0) ? use_woodbury_in : (k > 60);                              /* :863 */
    for (int64_t i = 0; i < n * k; ++i) coef_out[i] = NAN;                                                      /* :864 */
    uint8_t *ones = NULL;
    const uint8_t *is_valid = is_valid_in;
    if (!is_valid) { ones = (uint8_t *)malloc((size_t)(n ? n : 1)); memset(ones, 1, (size_t)n); is_valid = ones; }

    int64_t min_periods_valid = min_periods, n_valid = 0; /* :881-891 */
    for (int64_t i = 0; i < n; ++i) {
        if (is_valid[i]) n_valid += 1;
        if (n_valid == min_periods) { min_periods_valid = i + 1; break; }
    }
    if (n < ((n_valid > min_periods) ? n_valid : min_periods)) { free(ones); return; } /* :893-900 */
    if (min_periods_valid < 1) { free(ones); return; } /* reference would panic on index -1 */

    int64_t *dq = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n + 1)); /* VecDeque of valid indices */
    int64_t dq_head = 0, dq_tail = 0;
    roll_state st;
    st.k = k; st.woodbury = use_woodbury;
    st.m = dalloc((size_t)k * k); st.xty = dalloc(k);
    for (int i = 0; i < k * k; ++i) st.m[i] = 0.0;
    for (int i = 0; i < k; ++i) st.xty[i] = 0.0;
    st.woodbury = 0; /* accumulate XtX first */
    for (int64_t i = 0; i < min_periods_valid; ++i) { /* :909-921 */
        if (is_valid[i]) {
            roll_rank1(&st, x + i * k, y[i], +1.0);
            if (dq_tail - dq_head != window_size) dq[dq_tail++] = i;
        }
    }
    if (alpha > 0.0) for (int i = 0; i < k; ++i) st.m[i * k + i] += alpha; /* :924-926 */
    if (use_woodbury) { /* :929-932 */
        double *inv = dalloc((size_t)k * k);
        orc_inv(st.m, k, 0, inv);
        memcpy(st.m, inv, sizeof(double) * k * k);
        free(inv);
        st.woodbury = 1;
    }
    double *ci = dalloc(k);
    roll_solve(&st, ci); /* :939-943 */
    for (int j = 0; j < k; ++j) coef_out[(min_periods_valid - 1) * k + j] = ci[j];

    if (null_policy == ORC_NULL_DROP || null_policy == ORC_NULL_DROP_ZERO || null_policy == ORC_NULL_DROP_Y_ZERO_X) {
        int saturated = (dq_tail - dq_head == window_size); /* :951 */
        for (int64_t i = min_periods_valid; i < n; ++i) {
            if (is_valid[i]) {
                if (saturated) {
                    int64_t i0 = dq[dq_head];
                    roll_update(&st, x + i * k, y[i], x + i0 * k, y[i0], 1);
                    dq_head++;
                } else {
                    roll_update(&st, x + i * k, y[i], NULL, 0.0, 0);
                }
                roll_solve(&st, ci);
                for (int j = 0; j < k; ++j) coef_out[i * k + j] = ci[j];
                dq[dq_tail++] = i;
                if (!saturated) saturated = (dq_tail - dq_head == window_size);
            } else {
                for (int j = 0; j < k; ++j) coef_out[i * k + j] = ci[j]; /* forward fill :984 */
            }
        }
    } else { /* drop_window and everything else, :987-1029 */
        for (int64_t i = min_periods_valid; i < n; ++i) {
            const int64_t i_start = (i >= window_size) ? i - window_size : 0; /* saturating_sub */
            const int v_i = is_valid[i], v_s = is_valid[i_start];
            int64_t n_valid_window = 0;
            for (int64_t q = i_start + 1; q <= i; ++q) n_valid_window += is_valid[q] ? 1 : 0;
            if (v_i) {
                if ((i >= window_size) && v_s) roll_update(&st, x + i * k, y[i], x + i_start * k, y[i_start], 1);
                else roll_update(&st, x + i * k, y[i], NULL, 0.0, 0);
                if (n_valid_window >= n_valid) roll_solve(&st, ci);
            } else if (v_s && !v_i && (i >= window_size)) {
                roll_subtract(&st, x + i_start * k, y[i_start]);
                if (n_valid_window >= n_valid) roll_solve(&st, ci);
            }
            for (int j = 0; j < k; ++j) coef_out[i * k + j] = ci[j];
        }
    }
    free(ci); free(st.m); free(st.xty); free(dq); free(ones);
}

/* ------------------------------------------------- marshalling / predictions */

/* ex.rs:22-63 construct_features_array: k contiguous columns -> row-major n x k */
void orc_construct_features(const double *const *cols, int64_t n, int k, double *x) {
    for (int j = 0; j < k; ++j) {
        const double *c = cols[j];
        for (int64_t i = 0; i < n; ++i) x[i * k + j] = c[i];
    }
}

/* ex.rs:183 features.dot(coefficients) */
void orc_predict_static(const double *x, const double *beta, int64_t n, int k, double *pred) {
    for (int64_t i = 0; i < n; ++i) {
        double s = 0.0;
        for (int j = 0; j < k; ++j) s += x[i * k + j] * beta[j];
        pred[i] = s;
    }
}

/* ex.rs:184 (features * coefficients).sum_axis(1) */
void orc_predict_dynamic(const double *x, const double *coef, int64_t n, int k, double *pred) {
    for (int64_t i = 0; i < n; ++i) {
        double s = 0.0;
        for (int j = 0; j < k; ++j) s += x[i * k + j] * coef[i * k + j];
        pred[i] = s;
    }
}

/* ------------------------------------------------------------ batched driver */

int orc_batched_least_squares(const double *y, const double *const *x_cols,
                              const double *weights, int64_t n_rows, int k,
                              const int64_t *offs, int64_t n_groups, int add_intercept,
                              const orc_ols_params *p, double *coef_out, double *pred_out,
                              double *resid_out, int n_threads) {
    (void)n_rows;
    const int kt = k + (add_intercept ? 1 : 0);
    int err = 0;
#ifdef _OPENMP
    if (n_threads <= 0) n_threads = omp_get_max_threads();
#pragma omp parallel for schedule(dynamic, 8) num_threads(n_threads)
#endif
    for (int64_t g = 0; g < n_groups; ++g) {
        const int64_t s = offs[g], n = offs[g + 1] - offs[g];
        double *xf = tls_alloc(3, (size_t)n * kt), *yf = tls_alloc(4, (size_t)n), *sw = NULL;
        double beta[256];
        double *bp = (kt <= 256) ? beta : dalloc(kt);
        /* Python pre-processing, ls.py:184-196: sqrt_w = w.sqrt(); target *= sqrt_w;
         * every feature (intercept LAST, :188) *= sqrt_w. */
        if (weights) {
            sw = tls_alloc(5, (size_t)n);
            for (int64_t i = 0; i < n; ++i) sw[i] = sqrt(weights[s + i]);
        }
        /* plugin marshalling, ex.rs:22-63 + :79-91 */
        for (int j = 0; j < k; ++j) {
            const double *c = x_cols[j] + s;
            if (sw) for (int64_t i = 0; i < n; ++i) xf[i * kt + j] = c[i] * sw[i];
            else    for (int64_t i = 0; i < n; ++i) xf[i * kt + j] = c[i];
        }
        if (add_intercept) for (int64_t i = 0; i < n; ++i) xf[i * kt + k] = sw ? 1.0 * sw[i] : 1.0;
        for (int64_t i = 0; i < n; ++i) yf[i] = sw ? y[s + i] * sw[i] : y[s + i];
        int rc = orc_get_coefficients(yf, xf, n, kt, p, bp); /* ex.rs:395-396 */
        if (rc < 0) {
#ifdef _OPENMP
#pragma omp critical
#endif
            err = rc;
        }
        if (coef_out) for (int j = 0; j < kt; ++j) coef_out[g * kt + j] = bp[j];
        if (pred_out || resid_out) {
            for (int64_t i = 0; i < n; ++i) { /* ex.rs:398-405 make_predictions(x_fit, coef) */
                double a = 0.0;
                for (int j = 0; j < kt; ++j) a += xf[i * kt + j] * bp[j];
                if (sw) a *= 1.0 / sw[i];                        /* ls.py:234-235 */
                if (pred_out) pred_out[s + i] = a;
                if (resid_out) resid_out[s + i] = y[s + i] - a;  /* ls.py:239 original target */
            }
        }
        if (bp != beta) free(bp);
    }
    return err;
}

/* The same driver with groups dealt to the threads in contiguous static ranges and no residual output: the CPU-baseline
 * harness (orc_bench_static) calls this one, so that a thread re-visits the rows it first-touched. */
int orc_batched_least_squares_static(const double *y, const double *const *x_cols, const double *weights, int64_t n_rows, int k,
                                     const int64_t *offs, int64_t n_groups, int add_intercept, const orc_ols_params *p,
                                     double *coef_out, double *pred_out, int n_threads) {
    (void)n_rows;
    const int kt = k + (add_intercept ? 1 : 0);
    int err = 0;
#ifdef _OPENMP
    if (n_threads <= 0) n_threads = omp_get_max_threads();
#pragma omp parallel for schedule(static) num_threads(n_threads)
#endif
    for (int64_t g = 0; g < n_groups; ++g) {
        const int64_t s = offs[g], n = offs[g + 1] - offs[g];
        double *xf = tls_alloc(3, (size_t)n * kt), *yf = tls_alloc(4, (size_t)n), *sw = NULL;
        double beta[256];
        if (kt > 256) continue;
        if (weights) {
            sw = tls_alloc(5, (size_t)n);
            for (int64_t i = 0; i < n; ++i) sw[i] = sqrt(weights[s + i]);
        }
        for (int j = 0; j < k; ++j) {                      /* plugin marshalling, ex.rs:22-63 */
            const double *c = x_cols[j] + s;
            if (sw) for (int64_t i = 0; i < n; ++i) xf[i * kt + j] = c[i] * sw[i];
            else    for (int64_t i = 0; i < n; ++i) xf[i * kt + j] = c[i];
        }
        if (add_intercept) for (int64_t i = 0; i < n; ++i) xf[i * kt + k] = sw ? sw[i] : 1.0;
        for (int64_t i = 0; i < n; ++i) yf[i] = sw ? y[s + i] * sw[i] : y[s + i];
        if (orc_get_coefficients(yf, xf, n, kt, p, beta) < 0) err = -1;
        if (coef_out) for (int j = 0; j < kt; ++j) coef_out[g * kt + j] = beta[j];
        if (pred_out) {
            for (int64_t i = 0; i < n; ++i) {
                double a = 0.0;
                for (int j = 0; j < kt; ++j) a += xf[i * kt + j] * beta[j];
                if (sw) a *= 1.0 / sw[i];
                pred_out[s + i] = a;
            }
        }
    }
    return err;
}

int orc_batched_rls(const double *y, const double *const *x_cols, int64_t n_rows, int k,
                    const int64_t *offs, int64_t n_groups, int has_half_life, double half_life,
                    double initial_state_covariance, const double *mean0, const uint8_t *valid,
                    double *coef_out, double *pred_out, int n_threads) {
    (void)n_rows;
#ifdef _OPENMP
    if (n_threads <= 0) n_threads = omp_get_max_threads();
#pragma omp parallel for schedule(dynamic, 1) num_threads(n_threads)
#endif
    for (int64_t g = 0; g < n_groups; ++g) {
        const int64_t s = offs[g], n = offs[g + 1] - offs[g];
        double *xf = dalloc((size_t)n * k), *cf = coef_out ? coef_out + s * k : dalloc((size_t)n * k);
        const double *cols[256];
        for (int j = 0; j < k; ++j) cols[j] = x_cols[j] + s;
        orc_construct_features(cols, n, k, xf);
        orc_solve_rls(y + s, xf, n, k, has_half_life, half_life, initial_state_covariance, mean0,
                      valid ? valid + s : NULL, cf);
        if (pred_out) orc_predict_dynamic(xf, cf, n, k, pred_out + s); /* ex.rs:640-645 */
        if (!coef_out) free(cf);
        free(xf);
    }
    return 0;
}

int orc_batched_rolling(const double *y, const double *const *x_cols, int64_t n_rows, int k,
                        const int64_t *offs, int64_t n_groups, int64_t window_size,
                        int64_t min_periods, int use_woodbury, double alpha, int null_policy,
                        const uint8_t *valid, double *coef_out, double *pred_out, int n_threads) {
    (void)n_rows;
#ifdef _OPENMP
    if (n_threads <= 0) n_threads = omp_get_max_threads();
#pragma omp parallel for schedule(dynamic, 1) num_threads(n_threads)
#endif
    for (int64_t g = 0; g < n_groups; ++g) {
        const int64_t s = offs[g], n = offs[g + 1] - offs[g];
        double *xf = dalloc((size_t)n * k), *cf = coef_out ? coef_out + s * k : dalloc((size_t)n * k);
        const double *cols[256];
        for (int j = 0; j < k; ++j) cols[j] = x_cols[j] + s;
        orc_construct_features(cols, n, k, xf);
        orc_solve_rolling_ols(y + s, xf, n, k, window_size, min_periods, use_woodbury, alpha,
                              valid ? valid + s : NULL, null_policy, cf);
        if (pred_out) orc_predict_dynamic(xf, cf, n, k, pred_out + s); /* ex.rs:695-700 */
        if (!coef_out) free(cf);
        free(xf);
    }
    return 0;
}

/* --------------------------------------------------------------- statistics */

/* st.rs:15-37 */
void orc_residual_metrics_compute(const double *y, const double *pred, int64_t n,
                                  orc_residual_metrics *out) {
    double mean = 0.0;
    for (int64_t i = 0; i < n; ++i) mean += y[i];
    mean = n ? mean / (double)n : 0.0;
    double sse = 0.0, sae = 0.0, sst = 0.0;
    for (int64_t i = 0; i < n; ++i) {
        double e = y[i] - pred[i];
        sse += e * e; sae += fabs(e); sst += (y[i] - mean) * (y[i] - mean);
    }
    out->mse = sse / (double)n; out->mae = sae / (double)n; out->r2 = 1.0 - sse / sst;
}

/* continued fraction for the regularised incomplete beta function */
static double betacf(double a, double b, double x) {
    const double tiny = 1e-300;
    double qab = a + b, qap = a + 1.0, qam = a - 1.0, c = 1.0, d = 1.0 - qab * x / qap;
    if (fabs(d) < tiny) d = tiny;
    d = 1.0 / d;
    double h = d;
    for (int m = 1; m <= 500; ++m) {
        int m2 = 2 * m;
        double aa = m * (b - m) * x / ((qam + m2) * (a + m2));
        d = 1.0 + aa * d; if (fabs(d) < tiny) d = tiny;
        c = 1.0 + aa / c; if (fabs(c) < tiny) c = tiny;
        d = 1.0 / d; h *= d * c;
        aa = -(a + m) * (qab + m) * x / ((a + m2) * (qap + m2));
        d = 1.0 + aa * d; if (fabs(d) < tiny) d = tiny;
        c = 1.0 + aa / c; if (fabs(c) < tiny) c = tiny;
        d = 1.0 / d;
        double del = d * c;
        h *= del;
        if (fabs(del - 1.0) < 1e-16) break;
    }
    return h;
}

static double betai(double a, double b, double x) {
    if (x <= 0.0) return 0.0;
    if (x >= 1.0) return 1.0;
    double bt = exp(lgamma(a + b) - lgamma(a) - lgamma(b) + a * log(x) + b * log(1.0 - x));
    if (x < (a + 1.0) / (a + b + 2.0)) return bt * betacf(a, b, x) / a;
    return 1.0 - bt * betacf(b, a, 1.0 - x) / b;
}

/* st.rs:45-49: 2 * (1 - StudentsT(0,1,df).cdf(|t|)) == I_{df/(df+t^2)}(df/2, 1/2) */
double orc_student_t_two_sided_p(double t, double df) {
    if (isnan(t) || isnan(df)) return NAN;
    return betai(0.5 * df, 0.5, df / (df + t * t));
}

/* st.rs:79-156 compute_feature_metrics */
int orc_feature_metrics(const double *x, const double *y, int64_t n, int k, double lambda,
                        double *se, double *tv, double *pv) {
    double *xtx = dalloc((size_t)k * k), *xty = dalloc(k), *inv = dalloc((size_t)k * k), *coef = dalloc(k);
    for (int a = 0; a < k; ++a) {
        for (int b = 0; b < k; ++b) {
            double s = 0.0;
            for (int64_t i = 0; i < n; ++i) s += x[i * k + a] * x[i * k + b];
            xtx[a * k + b] = s + ((a == b) ? lambda : 0.0);
        }
        double s = 0.0;
        for (int64_t i = 0; i < n; ++i) s += x[i * k + a] * y[i];
        xty[a] = s;
    }
    double *l = dalloc((size_t)k * k);
    memcpy(l, xtx, sizeof(double) * k * k);
    if (chol_factor(l, k) != 0) { /* :101-111 */
        for (int j = 0; j < k; ++j) se[j] = tv[j] = pv[j] = NAN;
        free(xtx); free(xty); free(inv); free(coef); free(l);
        return 1;
    }
    free(l);
    orc_inv(xtx, k, 1, inv);
    for (int i = 0; i < k; ++i) { double a = 0.0; for (int j = 0; j < k; ++j) a += inv[i * k + j] * xty[j]; coef[i] = a; }
    double rss = 0.0;
    for (int64_t i = 0; i < n; ++i) {
        double a = 0.0;
        for (int j = 0; j < k; ++j) a += x[i * k + j] * coef[j];
        rss += (y[i] - a) * (y[i] - a);
    }
    double tr = 0.0;
    for (int i = 0; i < k; ++i) tr += inv[i * k + i];
    const double df = (lambda > 0.0) ? (double)n - tr : (double)n - (double)k; /* :124-128 */
    const double sigma2 = rss / df;
    for (int j = 0; j < k; ++j) {
        se[j] = sqrt(sigma2 * fabs(inv[j * k + j]));
        tv[j] = coef[j] / se[j];
        pv[j] = orc_student_t_two_sided_p(tv[j], df);
    }
    free(xtx); free(xty); free(inv); free(coef);
    return 0;
}


/* ---- CPU baseline harness (bench.py's cpu_baseline leg) --------------------------------------------------------------------
 * `passes` repetitions of the grouped static path with EVERY buffer allocated and first-touched (by the thread that will use it)
 * before the clock starts, groups dealt to threads in contiguous static ranges -- what the timed region holds is the reference's
 * per-group work only: [marshal ->] dispatch -> solve -> predictions.  solve_only = 1 times _get_least_squares_coefficients +
 * make_predictions on matrices marshalled (and sqrt(w)-scaled) beforehand: the variant that flatters the reference.
 * Returns wall seconds for all passes (< 0: the reference would have panicked). */
double orc_bench_static(const double *y, const double *const *x_cols, const double *weights, int64_t n_rows, int k,
                        const int64_t *offs, int64_t n_groups, int add_intercept, const orc_ols_params *p,
                        int solve_only, int passes, int n_threads) {
    const int kt = k + (add_intercept ? 1 : 0);
    double *pred = dalloc((size_t)n_rows), *coef = dalloc((size_t)n_groups * kt);
    double *xf_all = solve_only ? dalloc((size_t)n_rows * kt) : NULL, *yf_all = solve_only ? dalloc((size_t)n_rows) : NULL;
    int err = 0;
#ifdef _OPENMP
    if (n_threads <= 0) n_threads = omp_get_max_threads();
#else
    n_threads = 1;
#endif
    /* first touch + (solve_only) the marshalling pass, outside the clock */
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(n_threads)
#endif
    for (int64_t g = 0; g < n_groups; ++g) {
        const int64_t s = offs[g], n = offs[g + 1] - offs[g];
        for (int64_t i = 0; i < n; ++i) pred[s + i] = 0.0;
        for (int j = 0; j < kt; ++j) coef[g * kt + j] = 0.0;
        if (solve_only) {
            double *xf = xf_all + s * kt, *yf = yf_all + s;
            for (int64_t i = 0; i < n; ++i) {
                const double sw = weights ? sqrt(weights[s + i]) : 1.0;
                for (int j = 0; j < k; ++j) xf[i * kt + j] = x_cols[j][s + i] * sw;
                if (add_intercept) xf[i * kt + k] = sw;
                yf[i] = y[s + i] * sw;
            }
        }
    }
    double t0 = 0.0, t1 = 0.0;
#ifdef _OPENMP
    t0 = omp_get_wtime();
#endif
    for (int pass = 0; pass < passes; ++pass) {
        if (!solve_only) {
            const int rc = orc_batched_least_squares_static(y, x_cols, weights, n_rows, k, offs, n_groups, add_intercept, p, coef, pred, n_threads);
            if (rc < 0) err = rc;
        } else {
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(n_threads)
#endif
            for (int64_t g = 0; g < n_groups; ++g) {
                const int64_t s = offs[g], n = offs[g + 1] - offs[g];
                const double *xf = xf_all + s * kt;
                double beta[256];
                if (kt > 256) continue;
                if (orc_get_coefficients(yf_all + s, xf, n, kt, p, beta) < 0) err = -1;
                for (int j = 0; j < kt; ++j) coef[g * kt + j] = beta[j];
                for (int64_t i = 0; i < n; ++i) {
                    double a = 0.0;
                    for (int j = 0; j < kt; ++j) a += xf[i * kt + j] * beta[j];
                    pred[s + i] = a;
                }
            }
        }
    }
#ifdef _OPENMP
    t1 = omp_get_wtime();
#endif
    free(pred); free(coef); free(xf_all); free(yf_all);
    return err < 0 ? -1.0 : t1 - t0;
}

/* One sequence of the dynamic models, `passes` times, buffers pre-allocated: kind 0 = solve_recursive_least_squares + dynamic
 * make_predictions, kind 1 = solve_rolling_ols (drop-family deque) + predictions.  The reference runs a sequence on ONE core. */
double orc_bench_dynamic(int kind, const double *y, const double *const *x_cols, int64_t n, int k, double half_life,
                         int64_t window, int64_t min_periods, int passes) {
    double *xf = dalloc((size_t)n * k), *cf = dalloc((size_t)n * k), *pr = dalloc((size_t)n);
    orc_construct_features(x_cols, n, k, xf);
    for (int64_t i = 0; i < n * k; ++i) cf[i] = 0.0;
    for (int64_t i = 0; i < n; ++i) pr[i] = 0.0;
    double t0 = 0.0, t1 = 0.0;
#ifdef _OPENMP
    t0 = omp_get_wtime();
#endif
    for (int pass = 0; pass < passes; ++pass) {
        if (kind == 0) orc_solve_rls(y, xf, n, k, 1, half_life, 10.0, NULL, NULL, cf);
        else orc_solve_rolling_ols(y, xf, n, k, window, min_periods, -1, 0.0, NULL, ORC_NULL_DROP, cf);
        orc_predict_dynamic(xf, cf, n, k, pr);
    }
#ifdef _OPENMP
    t1 = omp_get_wtime();
#endif
    free(xf); free(cf); free(pr);
    return t1 - t0;
}
