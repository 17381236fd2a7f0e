/* CPU ORACLE (plain C) — TEST INFRASTRUCTURE ONLY; never linked into the product library.
 *
 * Dependency-free restatement of the AutoGP.jl hot path for one particle, driven by the same
 * postfix program as the C ABI (include/autogp_hip.h):
 *   leaves       src/GP.jl:135,161,194-197,236-239,279-283,324-329   (scalar eval_cov forms)
 *   combinators  src/GP.jl:371-373,417-419,485-491
 *   assembly     src/GP.jl:674-684  (compute_cov_matrix, the scalar twin of 666-668)
 *   likelihood   src/Model.jl:134-136 (log N(xs; 0, K) through a Cholesky factor)
 * PARITY UNPINNED BY THE REFERENCE: no Julia here and no numeric golden values upstream; this file
 * is pinned against oracle/oracle.py and oracle/oracle_mp.py (tests/test_oracle.py).
 *
 * ChangePoint follows the VECTORISED association (sig_i*sig_j)*k (src/GP.jl:494-501), the form the
 * model actually calls.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>

enum { OP_WN, OP_CONST, OP_LIN, OP_SE, OP_GE, OP_PER, OP_PLUS, OP_TIMES, OP_CP };

static double eval_program(const uint8_t* ops, int n_ops, const double* prm, double t1, double t2) {
  double st[256];
  int sp = 0, q = 0;
  const double pi = 3.14159265358979323846;
  for (int ip = 0; ip < n_ops; ++ip) {
    switch (ops[ip]) {
      case OP_WN: st[sp++] = (t1 == t2 ? 1.0 : 0.0) * prm[q]; q += 1; break;
      case OP_CONST: st[sp++] = prm[q]; q += 1; break;
      case OP_LIN: st[sp++] = prm[q + 1] + prm[q + 2] * ((t1 - prm[q]) * (t2 - prm[q])); q += 3; break;
      case OP_SE: { double dx = t1 - t2; st[sp++] = prm[q + 1] * exp(-.5 * dx * dx / (prm[q] * prm[q])); q += 2; break; }
      case OP_GE: { double dt = fabs(t1 - t2); st[sp++] = prm[q + 2] * exp(-pow(dt / prm[q], prm[q + 1])); q += 3; break; }
      case OP_PER: {
        double freq = pi / prm[q + 1], dx = fabs(t1 - t2), s = sin(freq * dx);
        st[sp++] = prm[q + 2] * exp((-2.0 / (prm[q] * prm[q])) * (s * s)); q += 3; break; }
      case OP_PLUS: sp--; st[sp - 1] = st[sp - 1] + st[sp]; break;
      case OP_TIMES: sp--; st[sp - 1] = st[sp - 1] * st[sp]; break;
      case OP_CP: {
        double loc = prm[q], sc = prm[q + 1]; q += 2;
        double s1 = .5 * (1.0 + tanh((loc - t1) / sc)), s2 = .5 * (1.0 + tanh((loc - t2) / sc));
        sp--;
        st[sp - 1] = (s1 * s2) * st[sp - 1] + ((1.0 - s1) * (1.0 - s2)) * st[sp];
        break; }
      default: return NAN;
    }
  }
  return st[0];
}

/* K (n x n, column-major, full) = eval_cov + noise I */
void agp_oracle_cov(const uint8_t* ops, int n_ops, const double* prm, double noise, const double* ts, int n,
                    double* K) {
  for (int j = 0; j < n; ++j)
    for (int i = j; i < n; ++i) {
      double v = eval_program(ops, n_ops, prm, ts[i], ts[j]);
      if (i == j) v += noise;
      K[(size_t)j * n + i] = v;
      K[(size_t)i * n + j] = v;
    }
}

/* Lower triangle only (column-major, upper part untouched): the input LAPACK dpotrf(uplo = 'L') wants.
 * Used by oracle/fast.py (C assembly + LAPACK factorisation), the large-n checker and the bench's CPU baseline. */
void agp_oracle_cov_lower(const uint8_t* ops, int n_ops, const double* prm, double noise, const double* ts, int n,
                          double* K) {
  for (int j = 0; j < n; ++j) {
    double* cj = K + (size_t)j * n;
    for (int i = j; i < n; ++i) cj[i] = eval_program(ops, n_ops, prm, ts[i], ts[j]);
    cj[j] += noise;
  }
}

/* in-place lower Cholesky, column-major; returns LAPACK-style info */
static int cholesky_lower(double* A, int n) {
  for (int j = 0; j < n; ++j) {
    double* cj = A + (size_t)j * n;
    for (int k = 0; k < j; ++k) {
      const double* ck = A + (size_t)k * n;
      const double ljk = ck[j];
      for (int i = j; i < n; ++i) cj[i] -= ck[i] * ljk;
    }
    if (!(cj[j] > 0.0)) return j + 1;
    const double d = sqrt(cj[j]);
    cj[j] = d;
    for (int i = j + 1; i < n; ++i) cj[i] /= d;
  }
  return 0;
}

double agp_oracle_logpdf(const uint8_t* ops, int n_ops, const double* prm, double noise, const double* ts,
                         const double* xs, int n, int* info) {
  *info = 0;
  if (n == 0) return 0.0;
  double* K = (double*)malloc(sizeof(double) * (size_t)n * n);
  double* a = (double*)malloc(sizeof(double) * (size_t)n);
  agp_oracle_cov(ops, n_ops, prm, noise, ts, n, K);
  *info = cholesky_lower(K, n);
  double lp = NAN;
  if (*info == 0) {
    double logdet = 0.0, q = 0.0;
    for (int i = 0; i < n; ++i) {
      double s = xs[i];
      for (int k = 0; k < i; ++k) s -= K[(size_t)k * n + i] * a[k];
      a[i] = s / K[(size_t)i * n + i];
      logdet += 2.0 * log(K[(size_t)i * n + i]);
      q += a[i] * a[i];
    }
    lp = -0.5 * (n * log(2.0 * 3.14159265358979323846) + logdet + q);
  }
  free(K); free(a);
  return lp;
}
