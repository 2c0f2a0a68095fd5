/*
 * oracle/letkf_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT.
 *
 * CPU restatement in plain C of the reference's LETKF analysis hot path
 * (see letkf_oracle.h).  The operation order of every routine follows the
 * cited Fortran so that results agree with the compiled reference
 * (oracle/_ref) to a few ulp; tests/test_oracle_vs_ref.py pins that.
 * Citations are file:line under /root/reference.
 */
#include "letkf_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define CM(a, ld, i, j) ((a)[(size_t)(i) + (size_t)(ld) * (size_t)(j)])

static inline double dsign(double a, double b) { return (b >= 0.0) ? fabs(a) : -fabs(a); }

/* common/netlib.f:504-523 -- sqrt(a^2+b^2) by the Moler-Morrison iteration */
double orc_pythag(double a, double b) {
  double p = fmax(fabs(a), fabs(b));
  if (p == 0.0) return p;
  double q = fmin(fabs(a), fabs(b)) / p;
  double r = q * q;
  for (;;) {
    double t = 4.0 + r;
    if (t == 4.0) break;
    double s = r / t;
    double u = 1.0 + 2.0 * s;
    p = u * p;
    double su = s / u;
    r = su * su * r;
  }
  return p;
}

/* common/netlib.f:1023-1186 -- Householder reduction to tridiagonal form with
 * accumulation of the transformation (Martin, Reinsch, Wilkinson 1968).
 * 1-based loop variables are kept (shifted at the access) so the control flow
 * can be checked line by line against the Fortran. */
void orc_tred2(int nm, int n, const double *a, double *d, double *e, double *z) {
#define Z(i, j) CM(z, nm, (i)-1, (j)-1)
#define A(i, j) CM(a, nm, (i)-1, (j)-1)
#define D(i) d[(i)-1]
#define E(i) e[(i)-1]
  int i, j, k, l;
  double f, g, h, hh, scale;

  for (i = 1; i <= n; ++i) {               /* :1067-1073 */
    for (j = i; j <= n; ++j) Z(j, i) = A(j, i);
    D(i) = A(n, i);
  }
  if (n > 1) {
    for (i = n; i >= 2; --i) {             /* :1077 */
      l = i - 1;
      h = 0.0;
      scale = 0.0;
      int skip = (l < 2);
      if (!skip) {
        for (k = 1; k <= l; ++k) scale += fabs(D(k));   /* :1084-1085 */
        if (scale == 0.0) skip = 1;
      }
      if (skip) {                          /* :1088-1096 */
        E(i) = D(l);
        for (j = 1; j <= l; ++j) {
          D(j) = Z(l, j);
          Z(i, j) = 0.0;
          Z(j, i) = 0.0;
        }
      } else {
        for (k = 1; k <= l; ++k) {         /* :1098-1101 */
          D(k) /= scale;
          h += D(k) * D(k);
        }
        f = D(l);
        g = -dsign(sqrt(h), f);
        E(i) = scale * g;
        h -= f * g;
        D(l) = f - g;
        for (j = 1; j <= l; ++j) E(j) = 0.0;            /* form a*u :1109 */
        for (j = 1; j <= l; ++j) {
          f = D(j);
          Z(j, i) = f;
          g = E(j) + Z(j, j) * f;
          for (k = j + 1; k <= l; ++k) {
            g += Z(k, j) * D(k);
            E(k) += Z(k, j) * f;
          }
          E(j) = g;
        }
        f = 0.0;                            /* form p :1127 */
        for (j = 1; j <= l; ++j) {
          E(j) /= h;
          f += E(j) * D(j);
        }
        hh = f / (h + h);
        for (j = 1; j <= l; ++j) E(j) -= hh * D(j);     /* form q :1136 */
        for (j = 1; j <= l; ++j) {          /* form reduced a :1139 */
          f = D(j);
          g = E(j);
          for (k = j; k <= l; ++k) Z(k, j) = Z(k, j) - f * E(k) - g * D(k);
          D(j) = Z(l, j);
          Z(i, j) = 0.0;
        }
      }
      D(i) = h;                             /* :1150 */
    }
    for (i = 2; i <= n; ++i) {              /* accumulation :1153 */
      l = i - 1;
      Z(n, l) = Z(l, l);
      Z(l, l) = 1.0;
      h = D(i);
      if (h != 0.0) {
        for (k = 1; k <= l; ++k) D(k) = Z(k, i) / h;
        for (j = 1; j <= l; ++j) {
          g = 0.0;
          for (k = 1; k <= l; ++k) g += Z(k, i) * Z(k, j);
          for (k = 1; k <= l; ++k) Z(k, j) -= g * D(k);
        }
      }
      for (k = 1; k <= l; ++k) Z(k, i) = 0.0;
    }
  }
  for (i = 1; i <= n; ++i) {                /* :1178-1181 */
    D(i) = Z(n, i);
    Z(n, i) = 0.0;
  }
  Z(n, n) = 1.0;
  E(1) = 0.0;
#undef A
}

/* common/netlib.f:718-887 -- implicit-shift QL on the tridiagonal matrix with
 * accumulation into z; at most 30 iterations per eigenvalue; ascending order. */
void orc_tql2(int nm, int n, double *d, double *e, double *z, int *ierr) {
  int i, j, k, l, m, l1, l2;
  double c, c2, c3 = 0.0, dl1, el1, f, g, h, p, r, s, s2 = 0.0, tst1, tst2;

  *ierr = 0;
  if (n == 1) return;
  for (i = 2; i <= n; ++i) E(i - 1) = E(i);
  f = 0.0;
  tst1 = 0.0;
  E(n) = 0.0;

  for (l = 1; l <= n; ++l) {
    j = 0;
    h = fabs(D(l)) + fabs(E(l));
    if (tst1 < h) tst1 = h;
    for (m = l; m <= n; ++m) {              /* :795-800 */
      tst2 = tst1 + fabs(E(m));
      if (tst2 == tst1) break;
    }
    if (m != l) {
      do {
        if (j == 30) {                      /* :803, :885 */
          *ierr = l;
          return;
        }
        ++j;
        l1 = l + 1;
        l2 = l1 + 1;
        g = D(l);
        p = (D(l1) - g) / (2.0 * E(l));
        r = orc_pythag(p, 1.0);
        D(l) = E(l) / (p + dsign(r, p));
        D(l1) = E(l) * (p + dsign(r, p));
        dl1 = D(l1);
        h = g - D(l);
        for (i = l2; i <= n; ++i) D(i) -= h;
        f += h;
        p = D(m);                           /* QL transformation :822 */
        c = 1.0;
        c2 = c;
        el1 = E(l1);
        s = 0.0;
        for (i = m - 1; i >= l; --i) {
          c3 = c2;
          c2 = c;
          s2 = s;
          g = c * E(i);
          h = c * p;
          r = orc_pythag(p, E(i));
          E(i + 1) = s * r;
          s = E(i) / r;
          c = p / r;
          p = c * D(i) - s * g;
          D(i + 1) = h + s * (c * g + s * D(i));
          for (k = 1; k <= n; ++k) {        /* form vector :843-847 */
            h = Z(k, i + 1);
            Z(k, i + 1) = s * Z(k, i) + c * h;
            Z(k, i) = c * Z(k, i) - s * h;
          }
        }
        p = -s * s2 * c3 * el1 * E(l) / dl1;
        E(l) = s * p;
        D(l) = c * p;
        tst2 = tst1 + fabs(E(l));
      } while (tst2 > tst1);
    }
    D(l) += f;                              /* :856 */
  }
  for (int ii = 2; ii <= n; ++ii) {         /* selection sort :859-880 */
    i = ii - 1;
    k = i;
    p = D(i);
    for (j = ii; j <= n; ++j) {
      if (D(j) >= p) continue;
      k = j;
      p = D(j);
    }
    if (k == i) continue;
    D(k) = D(i);
    D(i) = p;
    for (j = 1; j <= n; ++j) {
      p = Z(j, i);
      Z(j, i) = Z(j, k);
      Z(j, k) = p;
    }
  }
#undef Z
#undef D
#undef E
}

/* common/netlib.f:524-582, matz /= 0 branch */
void orc_rs(int nm, int n, const double *a, double *w, double *z, double *fv1, int *ierr) {
  if (n > nm) {
    *ierr = 10 * n;
    return;
  }
  orc_tred2(nm, n, a, w, fv1, z);
  orc_tql2(nm, n, w, fv1, z, ierr);
}

/* common/common_mtx.f90:41-99 */
int orc_mtx_eigen(int n, const double *a, double *eival, double *eivec, int *nrank_eff) {
  size_t nn = (size_t)n * (size_t)n;
  double *eivec8 = (double *)calloc(nn, sizeof(double));
  double *eival8 = (double *)malloc(sizeof(double) * (size_t)n);
  double *wrk1 = (double *)malloc(sizeof(double) * (size_t)n);
  int ierr = 0, rc = 0;

  orc_rs(n, n, a, eival8, eivec8, wrk1, &ierr);    /* :58-60 (a8 = a is read-only here) */
  if (ierr != 0) {                                   /* :61-64 */
    rc = 2;
    goto done;
  }
  int nr = n;
  if (eival8[n - 1] > 0) {                           /* :67-74 */
    for (int i = 0; i < n; ++i) {
      if (eival8[i] < fabs(eival8[n - 1]) * sqrt(DBL_EPSILON)) {
        --nr;
        eival8[i] = 0.0;
        for (int r = 0; r < n; ++r) CM(eivec8, n, r, i) = 0.0;
      }
    }
  } else {                                           /* :75-78 */
    rc = 2;
    goto done;
  }
  if (nr < n && eival8[0] != 0) {                    /* :80-91, incl. the eivec/eivec8 slip at :85 */
    int j = 0;
    for (int i = n; i >= 1; --i) {
      if (eival8[i - 1] == 0) {
        int src = n - nr - j; /* 1-based */
        eival8[i - 1] = eival8[src - 1];
        for (int r = 0; r < n; ++r) CM(eivec, n, r, i - 1) = CM(eivec8, n, r, src - 1);
        eival8[src - 1] = 0.0;
        for (int r = 0; r < n; ++r) CM(eivec8, n, r, src - 1) = 0.0;
        ++j;
      }
    }
  }
  for (int i = 0; i < n; ++i) {                      /* :93-96 descending */
    eival[i] = eival8[n - 1 - i];
    for (int r = 0; r < n; ++r) CM(eivec, n, r, i) = CM(eivec8, n, r, n - 1 - i);
  }
  if (nrank_eff) *nrank_eff = nr;
done:
  free(eivec8);
  free(eival8);
  free(wrk1);
  return rc;
}

/* common/netlibblas.f:491-506 ('T','N'), alpha=1 beta=0 */
static void gemm_tn(int m, int n, int k, const double *a, int lda, const double *b, int ldb,
                    double *c, int ldc) {
  for (int j = 0; j < n; ++j)
    for (int i = 0; i < m; ++i) {
      double temp = 0.0;
      for (int l = 0; l < k; ++l) temp += CM(a, lda, l, i) * CM(b, ldb, l, j);
      CM(c, ldc, i, j) = temp;
    }
}

/* common/netlibblas.f:510-530 ('N','T'), alpha=1 beta=0 */
static void gemm_nt(int m, int n, int k, const double *a, int lda, const double *b, int ldb,
                    double *c, int ldc) {
  for (int j = 0; j < n; ++j) {
    for (int i = 0; i < m; ++i) CM(c, ldc, i, j) = 0.0;
    for (int l = 0; l < k; ++l) {
      double temp = CM(b, ldb, j, l);
      if (temp != 0.0)
        for (int i = 0; i < m; ++i) CM(c, ldc, i, j) += temp * CM(a, lda, i, l);
    }
  }
}

/* common/common_letkf.f90:52-257 */
int orc_letkf_core(int ne, int nobs, int nobsl, const double *hdxb, const double *rdiag,
                   const double *rloc, const double *dep, double *parm_infl, double *trans,
                   double *transm, double *pao, const int *rdiag_wloc, const int *infl_update,
                   const double *depd, double *transmd) {
  const double sigma_b = 0.04;                       /* :79 */
  int wloc = rdiag_wloc ? (*rdiag_wloc != 0) : 0;    /* :84-87 */
  int iupd = infl_update ? (*infl_update != 0) : 0;
  size_t kk = (size_t)ne * (size_t)ne;

  if (nobsl == 0) {                                  /* :89-107 */
    memset(trans, 0, sizeof(double) * kk);
    for (int i = 0; i < ne; ++i) CM(trans, ne, i, i) = sqrt(*parm_infl);
    if (transm) memset(transm, 0, sizeof(double) * (size_t)ne);
    if (transmd) memset(transmd, 0, sizeof(double) * (size_t)ne);
    if (pao) {
      memset(pao, 0, sizeof(double) * kk);
      for (int i = 0; i < ne; ++i) CM(pao, ne, i, i) = *parm_infl / (double)(ne - 1);
    }
    return 0;
  }

  double *hdxb_rinv = (double *)malloc(sizeof(double) * (size_t)nobsl * (size_t)ne);
  double *hsub = (double *)malloc(sizeof(double) * (size_t)nobsl * (size_t)ne);
  double *eivec = (double *)malloc(sizeof(double) * kk);
  double *eival = (double *)malloc(sizeof(double) * (size_t)ne);
  double *pa = (double *)malloc(sizeof(double) * kk);
  double *work1 = (double *)malloc(sizeof(double) * kk);
  double *work2 = (double *)malloc(sizeof(double) * (size_t)ne * (size_t)nobsl);
  double *work3 = (double *)malloc(sizeof(double) * (size_t)ne);
  int rc = 0;

  for (int j = 0; j < ne; ++j)                       /* :111-123 */
    for (int i = 0; i < nobsl; ++i) {
      double v = CM(hdxb, nobs, i, j) / rdiag[i];
      if (!wloc) v = v * rloc[i];
      CM(hdxb_rinv, nobsl, i, j) = v;
      CM(hsub, nobsl, i, j) = CM(hdxb, nobs, i, j);  /* hdxb(1:nobsl,:) copy-in, :127 */
    }
  gemm_tn(ne, ne, nobsl, hdxb_rinv, nobsl, hsub, nobsl, work1, ne);   /* :127-128 */
  double rho = 1.0 / *parm_infl;                     /* :140-143 */
  for (int i = 0; i < ne; ++i) CM(work1, ne, i, i) += (double)(ne - 1) * rho;
  int nrank;
  rc = orc_mtx_eigen(ne, work1, eival, eivec, &nrank);     /* :147 */
  if (rc != 0) goto done;
  for (int j = 0; j < ne; ++j)                       /* :151-157 */
    for (int i = 0; i < ne; ++i) CM(work1, ne, i, j) = CM(eivec, ne, i, j) / eival[j];
  gemm_nt(ne, ne, ne, work1, ne, eivec, ne, pa, ne);
  gemm_nt(ne, nobsl, ne, pa, ne, hdxb_rinv, nobsl, work2, ne);         /* :169-170 */
  for (int i = 0; i < ne; ++i) {                     /* :182-187 */
    double t = CM(work2, ne, i, 0) * dep[0];
    for (int j = 1; j < nobsl; ++j) t += CM(work2, ne, i, j) * dep[j];
    work3[i] = t;
  }
  if (depd && transmd) {                             /* :188-195 */
    for (int i = 0; i < ne; ++i) {
      double t = CM(work2, ne, i, 0) * depd[0];
      for (int j = 1; j < nobsl; ++j) t += CM(work2, ne, i, j) * depd[j];
      transmd[i] = t;
    }
  }
  for (int j = 0; j < ne; ++j) {                     /* :199-206 */
    double r = sqrt((double)(ne - 1) / eival[j]);
    for (int i = 0; i < ne; ++i) CM(work1, ne, i, j) = CM(eivec, ne, i, j) * r;
  }
  gemm_nt(ne, ne, ne, work1, ne, eivec, ne, trans, ne);
  if (transm) {                                      /* :218-226 */
    memcpy(transm, work3, sizeof(double) * (size_t)ne);
  } else {
    for (int j = 0; j < ne; ++j)
      for (int i = 0; i < ne; ++i) CM(trans, ne, i, j) += work3[i];
  }
  if (pao) memcpy(pao, pa, sizeof(double) * kk);     /* :227 */

  if (iupd) {                                        /* :233-254 */
    double parm[4] = {0.0, 0.0, 0.0, 0.0};
    for (int i = 0; i < nobsl; ++i) {
      double t = dep[i] * dep[i] / rdiag[i];
      if (!wloc) t = t * rloc[i];
      parm[0] += t;
    }
    for (int j = 0; j < ne; ++j)
      for (int i = 0; i < nobsl; ++i) parm[1] += CM(hdxb_rinv, nobsl, i, j) * CM(hdxb, nobs, i, j);
    parm[1] = parm[1] / (double)(ne - 1);
    for (int i = 0; i < nobsl; ++i) parm[2] += rloc[i];
    parm[3] = (parm[0] - parm[2]) / parm[1] - *parm_infl;
    double t = (*parm_infl * parm[1] + parm[2]) / parm[1];
    double sigma_o = 2.0 / parm[2] * (t * t);
    double gain = sigma_b * sigma_b / (sigma_o + sigma_b * sigma_b);
    *parm_infl = *parm_infl + gain * parm[3];
  }
done:
  free(hdxb_rinv);
  free(hsub);
  free(eivec);
  free(eival);
  free(pa);
  free(work1);
  free(work2);
  free(work3);
  return rc;
}

/* scale/letkf/letkf_tools.f90:1911-1948 */
double orc_relax_beta(const orc_beta_params *bp, double ri, double rj, double rz) {
  double beta = 1.0;
  if (bp->radar_only && rz > bp->radar_zmax + bp->vert_local_radar * ORC_DIST_ZERO_FAC) return 0.0;
  if (bp->boundary_buffer_width > 0.0) {
    double di = fmin(ri - bp->ihalo, bp->nlong + bp->ihalo + 1 - ri) * bp->dx;
    double dj = fmin(rj - bp->jhalo, bp->nlatg + bp->jhalo + 1 - rj) * bp->dy;
    double dist_bdy = fmin(di, dj) / bp->boundary_buffer_width;
    if (dist_bdy < 1.0) beta = fmax(dist_bdy, 0.0);
  }
  return beta;
}

/* scale/letkf/letkf_tools.f90:139-157.  The reference indexes var_local with the CLASS number var_local_n2nc(i)
 * (:145) and var_local_n2n with var_local_n2nc(n) (:147); restated as written. */
void orc_var_local_classes(int nvar, int nlt, const double *var_local, int32_t *n2nc, int32_t *n2n, int32_t *nclass) {
  /* 1-based work arrays, exactly the reference's */
  int *c1 = (int *)calloc((size_t)nvar + 1, sizeof(int));
  int *r1 = (int *)calloc((size_t)nvar + 1, sizeof(int));
  int ncmax = 1;
  c1[1] = 1;
  r1[1] = 1;
  for (int n = 2; n <= nvar; ++n) {
    int found = 0;
    for (int i = 1; i <= ncmax; ++i) {
      double mx = 0.0;
      for (int t = 0; t < nlt; ++t) {
        double d = fabs(var_local[(c1[i] - 1) + (size_t)nvar * t] - var_local[(n - 1) + (size_t)nvar * t]);
        if (d > mx) mx = d;
      }
      if (mx < DBL_MIN) { /* tiny(var_local) */
        c1[n] = c1[i];
        r1[n] = r1[c1[n]];
        found = 1;
        break;
      }
    }
    if (!found) {
      ncmax = ncmax + 1;
      c1[n] = ncmax;
      r1[n] = n;
    }
  }
  for (int n = 1; n <= nvar; ++n) {
    n2nc[n - 1] = c1[n] - 1;
    n2n[n - 1] = r1[n] - 1;
  }
  *nclass = ncmax;
  free(c1);
  free(r1);
}

/* scale/letkf/letkf_tools.f90:167-192 */
void orc_ctype_merge(int nctype, const int32_t *elm_u_ctype, const int32_t *typ_ctype, int nid_obs, int nobtype,
                     const int32_t *ctype_merge, int32_t *n_merge, int32_t *ic_merge) {
  (void)nobtype;
#define CMRG(ic) ctype_merge[(elm_u_ctype[ic] - 1) + (size_t)nid_obs * (typ_ctype[ic] - 1)]
  for (int ic = 0; ic < nctype; ++ic) n_merge[ic] = 1;
  for (int ic = 0; ic < nctype; ++ic) {
    if (n_merge[ic] > 0) {
      ic_merge[(size_t)ic * nctype + 0] = ic;
      if (CMRG(ic) > 0) {
        for (int ic2 = ic + 1; ic2 < nctype; ++ic2) {
          if (CMRG(ic2) == CMRG(ic)) {
            n_merge[ic] = n_merge[ic] + 1;
            ic_merge[(size_t)ic * nctype + n_merge[ic] - 1] = ic2;
            n_merge[ic2] = 0;
          }
        }
      }
    }
  }
#undef CMRG
}

/* scale/letkf/letkf_tools.f90:197-203 */
int orc_radar_only(int nctype, const int32_t *typ_ctype, int typ_radar) {
  int radar_only = 1;
  for (int ic = 0; ic < nctype; ++ic) {
    if (typ_ctype[ic] != typ_radar) {
      radar_only = 0;
      break;
    }
  }
  return radar_only;
}

/* scale/letkf/letkf_tools.f90:237-267 (the read-in branch leaves the caller's field in place) */
void orc_infl_init(int64_t n, double *work3d, double infl_mul, double infl_mul_min) {
  if (infl_mul > 0.0)
    for (int64_t i = 0; i < n; ++i) work3d[i] = infl_mul;
  if (infl_mul_min > 0.0)
    for (int64_t i = 0; i < n; ++i) work3d[i] = work3d[i] > infl_mul_min ? work3d[i] : infl_mul_min;
}

/* scale/letkf/letkf_tools.f90:1953-1966 */
void orc_weight_rtpp(int k, double relax_alpha, const double *w, double infl, double *wrlx) {
  for (size_t i = 0; i < (size_t)k * (size_t)k; ++i) wrlx[i] = (1.0 - relax_alpha) * w[i];
  for (int m = 0; m < k; ++m) CM(wrlx, k, m, m) += relax_alpha * sqrt(infl);
}

/* scale/letkf/letkf_tools.f90:1971-2002 */
void orc_weight_rtps(int k, double relax_alpha_spread, const double *w, const double *pa,
                     const double *xb, double infl, double *wrlx, double *infl_out) {
  double var_g = 0.0, var_a = 0.0;
  for (int m = 0; m < k; ++m) {
    var_g += xb[m] * xb[m];
    for (int kk = 0; kk < k; ++kk) var_a += xb[kk] * CM(pa, k, kk, m) * xb[m];
  }
  size_t n2 = (size_t)k * (size_t)k;
  if (var_g > 0.0 && var_a > 0.0) {
    *infl_out = relax_alpha_spread * sqrt(var_g * infl / (var_a * (double)(k - 1))) -
                relax_alpha_spread + 1.0;
    for (size_t i = 0; i < n2; ++i) wrlx[i] = w[i] * (*infl_out);
  } else {
    for (size_t i = 0; i < n2; ++i) wrlx[i] = w[i];
    *infl_out = 1.0;
  }
}

/* scale/letkf/letkf_tools.f90:1793-1906 */
double orc_obs_local_cal(double ri, double rj, double rlev, double rz, double varloc,
                         int vmode, double hori_loc, double vert_loc, double rain_base,
                         double ob_ri, double ob_rj, double ob_lev, double ob_dat, double ob_err,
                         double dx, double dy, double *ndist, double *nrdiag) {
  double nrloc = 0.0, nd_v, nd_h;
  *nrdiag = -1.0;
  *ndist = -1.0;
  nrloc = varloc;                                    /* :1840 (nvar > 0) */
  if (nrloc < DBL_MIN) return 0.0;                   /* :1843 tiny(var_local) */
  if (vert_loc == 0.0) nd_v = 0.0;                   /* :1851-1865 */
  else if (vmode == 2) nd_v = fabs(log(ob_dat) - log(rlev)) / vert_loc;
  else if (vmode == 3) nd_v = fabs(log(rain_base) - log(rlev)) / vert_loc;
  else if (vmode == 1) nd_v = fabs(ob_lev - rz) / vert_loc;
  else nd_v = fabs(log(ob_lev) - log(rlev)) / vert_loc;
  if (nd_v > ORC_DIST_ZERO_FAC) return 0.0;          /* :1869 */
  double rdx = (ri - ob_ri) * dx;                    /* :1876-1878 */
  double rdy = (rj - ob_rj) * dy;
  nd_h = sqrt(rdx * rdx + rdy * rdy) / hori_loc;
  if (nd_h > ORC_DIST_ZERO_FAC) return 0.0;          /* :1881 */
  *ndist = nd_h * nd_h + nd_v * nd_v;                /* :1888 */
  if (*ndist > ORC_DIST_ZERO_FAC_SQUARE) {           /* :1891 */
    *ndist = -1.0;
    return 0.0;
  }
  nrloc = nrloc * exp(-0.5 * (*ndist));              /* :1899 */
  *nrdiag = ob_err * ob_err / nrloc;                 /* :1903 */
  return nrloc;
}

/* scale/common/common_scale.f90:1513-1552 */
void orc_ensmean(int k, int nv, int64_t npts, double *x, int64_t sp, int64_t sm, int64_t sv) {
#pragma omp parallel for schedule(static)
  for (int64_t pt = 0; pt < npts; ++pt)
    for (int v = 0; v < nv; ++v) {
      double *b = x + pt * sp + v * sv;
      double s = b[0];
      for (int m = 1; m < k; ++m) s += b[m * sm];
      b[k * sm] = s / (double)k;
    }
}

/* scale/letkf/letkf_tools.f90:209-230 */
void orc_to_perturbations(int k, int nv, int64_t npts, double *x, int64_t sp, int64_t sm, int64_t sv) {
#pragma omp parallel for schedule(static)
  for (int64_t pt = 0; pt < npts; ++pt)
    for (int v = 0; v < nv; ++v) {
      double *b = x + pt * sp + v * sv;
      double mean = b[k * sm];
      for (int m = 0; m < k; ++m) b[m * sm] -= mean;
    }
}

/* scale/letkf/letkf_tools.f90:313-527: one (ij, ilev) point, single
 * variable-localisation class (the namelist default), nv2d = 0. */
static int das_point(const orc_das_params *p, int64_t pt, int nobsl, const int32_t *idx,
                     const double *rdiag, const double *rloc, const double *ensval, int64_t kld,
                     const double *dep, double beta, double *infl, int64_t npts,
                     const double *gues, double *anal, int64_t sp, int64_t sm, int64_t sv,
                     double *trans_o, double *transm_o, double *pa_o,
                     double *hdxf, double *rd, double *rl, double *dp, double *dpd,
                     double *trans, double *transm, double *transmd, double *pa, double *wrlx, double *rtps_o) {
  const int k = p->k, nv = p->nv;
  const double *g0 = gues + pt * sp;
  double *a0 = anal + pt * sp;
  int rc = 0;

  const unsigned vmask = p->var_mask ? p->var_mask : ~0u;
  if (beta == 0.0) {                                 /* :333-359 */
    for (int v = 0; v < nv; ++v) {
      if (!((vmask >> v) & 1u)) continue;
      for (int m = 0; m < k; ++m) a0[m * sm + v * sv] = g0[k * sm + v * sv] + g0[m * sm + v * sv];
      if (p->det_run) a0[(k + 1) * sm + v * sv] = g0[(k + 1) * sm + v * sv];
    }
    return 0;
  }
  int done = 0;
  for (int v = 0; v < nv; ++v) {                     /* :366 */
    if (!((vmask >> v) & 1u)) continue;              /* another class: its own call */
    double *infl_v = infl + pt + npts * v;
    if (p->q_update_top > 0.0 && g0[k * sm + p->iv_p * sv] < p->q_update_top && v >= p->iv_q_first &&
        v <= p->iv_q_last) {                         /* :371-385 */
      for (int m = 0; m < k; ++m) a0[m * sm + v * sv] = g0[k * sm + v * sv] + g0[m * sm + v * sv];
      if (p->det_run) a0[(k + 1) * sm + v * sv] = g0[(k + 1) * sm + v * sv];
      continue;
    }
    double parm = p->relax_to_inflated_prior ? *infl_v : 1.0;   /* :387-391, read BEFORE the update */
    if (done) {                                      /* :394-406 */
      if (p->infl_adaptive) *infl_v = infl[pt + npts * (size_t)(done - 1)];
    } else {                                         /* :409-439 */
      for (int i = 0; i < nobsl; ++i) {              /* the obs_local copy-out :1463-1469 */
        const double *row = ensval + (int64_t)idx[i] * kld;
        for (int m = 0; m < k; ++m) CM(hdxf, nobsl > 0 ? nobsl : 1, i, m) = row[m];
        rd[i] = rdiag[i];
        rl[i] = rloc[i];
        dp[i] = dep[idx[i]];
        if (p->det_run) dpd[i] = row[k];            /* ensval(mmdetobs,iob) */
      }
      int one = 1, iu = p->infl_adaptive;
      rc = orc_letkf_core(k, nobsl > 0 ? nobsl : 1, nobsl, hdxf, rd, rl, dp, infl_v, trans, transm,
                          (p->relax_alpha_spread != 0.0) ? pa : NULL, &one, &iu,
                          p->det_run ? dpd : NULL, p->det_run ? transmd : NULL);
      if (rc != 0) return rc;
      done = v + 1;
      if (trans_o) memcpy(trans_o, trans, sizeof(double) * (size_t)k * (size_t)k);
      if (transm_o) memcpy(transm_o, transm, sizeof(double) * (size_t)k);
      if (pa_o && p->relax_alpha_spread != 0.0) memcpy(pa_o, pa, sizeof(double) * (size_t)k * (size_t)k);
    }
    if (p->relax_alpha != 0.0) {                     /* :457-469 */
      orc_weight_rtpp(k, p->relax_alpha, trans, parm, wrlx);
    } else if (p->relax_alpha_spread != 0.0) {
      double xb[k], tmpinfl;
      for (int m = 0; m < k; ++m) xb[m] = g0[m * sm + v * sv];
      orc_weight_rtps(k, p->relax_alpha_spread, trans, pa, xb, parm, wrlx, &tmpinfl);
      if (rtps_o) rtps_o[pt + npts * (size_t)v] = tmpinfl;               /* work3da(ij,ilev,n) :461-462 */
    } else {
      memcpy(wrlx, trans, sizeof(double) * (size_t)k * (size_t)k);
    }
    for (int m = 0; m < k; ++m) {                    /* :472-477 */
      for (int kk = 0; kk < k; ++kk) CM(wrlx, k, kk, m) = (CM(wrlx, k, kk, m) + transm[kk]) * beta;
      CM(wrlx, k, m, m) += (1.0 - beta);
    }
    for (int m = 0; m < k; ++m) {                    /* :480-486 */
      double t = g0[k * sm + v * sv];
      for (int kk = 0; kk < k; ++kk) t = t + g0[kk * sm + v * sv] * CM(wrlx, k, kk, m);
      a0[m * sm + v * sv] = t;
    }
    if (p->det_run) {                                /* :489-497 */
      double t = 0.0;
      for (int kk = 0; kk < k; ++kk) t = t + g0[kk * sm + v * sv] * transmd[kk];
      a0[(k + 1) * sm + v * sv] = g0[(k + 1) * sm + v * sv] + t * beta;
    }
    if (p->q_sprd_max > 0.0 && v == p->iv_q_first) { /* :500-513 */
      double q_mean = 0.0, q_sprd = 0.0, q_anal[k];
      for (int m = 0; m < k; ++m) q_mean += a0[m * sm + v * sv];
      q_mean /= (double)k;
      for (int m = 0; m < k; ++m) {
        q_anal[m] = a0[m * sm + v * sv] - q_mean;
        q_sprd += q_anal[m] * q_anal[m];
      }
      q_sprd = sqrt(q_sprd / (double)(k - 1)) / q_mean;
      if (q_sprd > p->q_sprd_max)
        for (int m = 0; m < k; ++m) a0[m * sm + v * sv] = q_mean + q_anal[m] * p->q_sprd_max / q_sprd;
    }
  }
  return rc;
}

int orc_das_letkf_points(const orc_das_params *p, int64_t npts, const int64_t *obs_off,
                         const int32_t *obs_idx, const double *rdiag_l, const double *rloc_l,
                         const double *ensval, int64_t kld, const double *dep, const double *beta,
                         double *infl, const double *gues, double *anal, int64_t sp, int64_t sm,
                         int64_t sv, double *trans_out, double *transm_out, double *pa_out,
                         int32_t *status) {
  return orc_das_letkf_points_diag(p, npts, obs_off, obs_idx, rdiag_l, rloc_l, ensval, kld, dep, beta, infl, gues,
                                   anal, sp, sm, sv, trans_out, transm_out, pa_out, status, NULL);
}

int orc_das_letkf_points_diag(const orc_das_params *p, int64_t npts, const int64_t *obs_off,
                              const int32_t *obs_idx, const double *rdiag_l, const double *rloc_l,
                              const double *ensval, int64_t kld, const double *dep, const double *beta,
                              double *infl, const double *gues, double *anal, int64_t sp, int64_t sm,
                              int64_t sv, double *trans_out, double *transm_out, double *pa_out,
                              int32_t *status, double *rtps_out) {
  if (rtps_out)
    for (int64_t e = 0; e < npts * p->nv; ++e) rtps_out[e] = 1.0;       /* work3da = 1.0d0 :274 */
  const int k = p->k;
  int64_t nmax = 1;
  for (int64_t pt = 0; pt < npts; ++pt) {
    int64_t n = obs_off[pt + 1] - obs_off[pt];
    if (n > nmax) nmax = n;
  }
  int worst = 0;
#ifdef _OPENMP
  int nth = p->nthreads > 0 ? p->nthreads : omp_get_max_threads();
#else
  int nth = 1;
#endif
  (void)nth;
#pragma omp parallel num_threads(nth)
  {
    /* per-thread buffers, as letkf_tools.f90:292-302 (sized to the largest
     * local list here instead of nobstotal) */
    double *hdxf = (double *)malloc(sizeof(double) * (size_t)nmax * (size_t)k);
    double *rd = (double *)malloc(sizeof(double) * (size_t)nmax * 4);
    double *rl = rd + nmax, *dp = rl + nmax, *dpd = dp + nmax;
    size_t kk = (size_t)k * (size_t)k;
    double *trans = (double *)malloc(sizeof(double) * (3 * kk + 2 * (size_t)k));
    double *pa = trans + kk, *wrlx = pa + kk, *transm = wrlx + kk, *transmd = transm + k;
    /* dynamic schedule with small chunks: letkf_tools.f90:290,319 */
#pragma omp for schedule(dynamic, 4)
    for (int64_t pt = 0; pt < npts; ++pt) {
      int64_t o = obs_off[pt];
      int nobsl = (int)(obs_off[pt + 1] - o);
      int rc = das_point(p, pt, nobsl, obs_idx + o, rdiag_l + o, rloc_l + o, ensval, kld, dep,
                         beta ? beta[pt] : 1.0, infl, npts, gues, anal, sp, sm, sv,
                         trans_out ? trans_out + (size_t)pt * kk : NULL,
                         transm_out ? transm_out + (size_t)pt * (size_t)k : NULL,
                         pa_out ? pa_out + (size_t)pt * kk : NULL, hdxf, rd, rl, dp, dpd, trans,
                         transm, transmd, pa, wrlx, rtps_out);
      if (status) status[pt] = rc;
      if (rc != 0) {
#pragma omp critical
        if (rc > worst) worst = rc;
      }
    }
    free(hdxf);
    free(rd);
    free(trans);
  }
  return worst;
}

/* The loop body at ilev = 1 WITH 2-D variables, scale/letkf/letkf_tools.f90:313-686, all variable-localisation
 * classes of a point in one pass as the reference has it: the 3-D loop (:366-527) and the 2-D loop (:530-659) share
 * trans_done(:) / trans(:,:,n2nc) / transm / transmd / pa, so a 2-D variable whose class was already solved for a 3-D
 * variable re-uses those weights and copies its adaptive inflation from work3d(ij,ilev,n2n) (:549-554), otherwise from
 * work2d(ij,n2n-nv3d) (:555-562) or it solves for itself with work2d(ij,n) (:566-601).  The 2-D variables know no
 * Q_UPDATE_TOP skip and no spread clamp.  (The reference is compiled with nv2d = 0, scale/common/common_scale.f90:54.)
 * Inputs: n2nc / n2n [nv3d + nv2d] = var_local_n2nc / var_local_n2n, 1-based as in the reference; the local lists of
 * class c (what obs_local returns for a variable of that class) are the CSR slice obs_off[c*(nij1+1) + ij] ..
 * [.. + ij + 1] of obs_idx / rdiag_l / rloc_l; work3d [nij1*nv3d] is the ilev = 1 slice, work2d [nij1*nv2d];
 * gues3 / anal3 element (ij, m, n) at ij*sp3 + m*sm3 + n*sv3 (again the ilev = 1 slice), gues2 / anal2 likewise. */
int orc_das_letkf_level1_2d(const orc_das_params *p, int nv2d, const int32_t *n2nc, const int32_t *n2n, int nclass,
                            int64_t nij1, const int64_t *obs_off, const int32_t *obs_idx, const double *rdiag_l,
                            const double *rloc_l, const double *ensval, int64_t kld, const double *dep,
                            const double *beta_a, double *work3d, double *work2d, const double *gues3, double *anal3,
                            int64_t sp3, int64_t sm3, int64_t sv3, const double *gues2, double *anal2, int64_t sp2,
                            int64_t sm2, int64_t sv2) {
  const int k = p->k, nv3d = p->nv;
  const size_t kk = (size_t)k * (size_t)k;
  int64_t nmax = 1;
  for (int c = 0; c < nclass; ++c)
    for (int64_t ij = 0; ij < nij1; ++ij) {
      const int64_t n = obs_off[(int64_t)c * (nij1 + 1) + ij + 1] - obs_off[(int64_t)c * (nij1 + 1) + ij];
      if (n > nmax) nmax = n;
    }
  double *hdxf = (double *)malloc(sizeof(double) * (size_t)nmax * (size_t)k);
  double *rd = (double *)malloc(sizeof(double) * (size_t)nmax * 4);
  double *rl = rd + nmax, *dp = rl + nmax, *dpd = dp + nmax;
  double *trans = (double *)malloc(sizeof(double) * ((2 * kk + 2 * (size_t)k) * (size_t)nclass + kk));   /* :297-302 */
  double *pa = trans + kk * nclass, *transm = pa + kk * nclass, *transmd = transm + (size_t)k * nclass;
  double *wrlx = transmd + (size_t)k * nclass;
  int *trans_done = (int *)malloc(sizeof(int) * (size_t)nclass);
  int worst = 0;
  for (int64_t ij = 0; ij < nij1; ++ij) {
    for (int c = 0; c < nclass; ++c) trans_done[c] = 0;                  /* :321 */
    const double beta = beta_a ? beta_a[ij] : 1.0;                       /* :325 */
    const double *g3 = gues3 + ij * sp3, *g2 = gues2 + ij * sp2;
    double *a3 = anal3 + ij * sp3, *a2 = anal2 + ij * sp2;
    if (beta == 0.0) {                                                   /* :333-359 */
      for (int n = 0; n < nv3d; ++n) {
        for (int m = 0; m < k; ++m) a3[m * sm3 + n * sv3] = g3[k * sm3 + n * sv3] + g3[m * sm3 + n * sv3];
        if (p->det_run) a3[(k + 1) * sm3 + n * sv3] = g3[(k + 1) * sm3 + n * sv3];
      }
      for (int n = 0; n < nv2d; ++n) {
        for (int m = 0; m < k; ++m) a2[m * sm2 + n * sv2] = g2[k * sm2 + n * sv2] + g2[m * sm2 + n * sv2];
        if (p->det_run) a2[(k + 1) * sm2 + n * sv2] = g2[(k + 1) * sm2 + n * sv2];
      }
      continue;
    }
    for (int nn = 0; nn < nv3d + nv2d; ++nn) {                           /* :366 DO n=1,nv3d ; :532 DO n=1,nv2d */
      const int is2d = nn >= nv3d, n = is2d ? nn - nv3d : nn;
      const int cls = n2nc[nn] - 1, rep = n2n[nn] - 1;                   /* :368-369, :534-535 */
      const double *g = is2d ? g2 : g3;
      double *a = is2d ? a2 : a3;
      const int64_t sm = is2d ? sm2 : sm3, sv = is2d ? sv2 : sv3;
      double *work = is2d ? work2d + ij + nij1 * n : work3d + ij + nij1 * n;
      if (!is2d && p->q_update_top > 0.0 && g3[k * sm3 + p->iv_p * sv3] < p->q_update_top && n >= p->iv_q_first &&
          n <= p->iv_q_last) {                                           /* :371-385 */
        for (int m = 0; m < k; ++m) a[m * sm + n * sv] = g[k * sm + n * sv] + g[m * sm + n * sv];
        if (p->det_run) a[(k + 1) * sm + n * sv] = g[(k + 1) * sm + n * sv];
        continue;
      }
      const double parm = p->relax_to_inflated_prior ? *work : 1.0;      /* :387-391, :537-541 */
      double *tr = trans + kk * cls, *tm = transm + (size_t)k * cls, *tmd = transmd + (size_t)k * cls, *pc = pa + kk * cls;
      if (trans_done[cls]) {                                             /* :394-406, :544-568 */
        if (p->infl_adaptive) *work = (rep < nv3d) ? work3d[ij + nij1 * rep] : work2d[ij + nij1 * (rep - nv3d)];
      } else {                                                           /* :409-439, :569-607 */
        const int64_t o = obs_off[(int64_t)cls * (nij1 + 1) + ij];
        const int nobsl = (int)(obs_off[(int64_t)cls * (nij1 + 1) + ij + 1] - o);
        for (int i = 0; i < nobsl; ++i) {
          const double *row = ensval + (int64_t)obs_idx[o + i] * kld;
          for (int m = 0; m < k; ++m) CM(hdxf, nobsl > 0 ? nobsl : 1, i, m) = row[m];
          rd[i] = rdiag_l[o + i];
          rl[i] = rloc_l[o + i];
          dp[i] = dep[obs_idx[o + i]];
          if (p->det_run) dpd[i] = row[k];
        }
        int one = 1, iu = p->infl_adaptive;
        const int rc = orc_letkf_core(k, nobsl > 0 ? nobsl : 1, nobsl, hdxf, rd, rl, dp, work, tr, tm,
                                      (p->relax_alpha_spread != 0.0) ? pc : NULL, &one, &iu,
                                      p->det_run ? dpd : NULL, p->det_run ? tmd : NULL);
        if (rc > worst) worst = rc;
        trans_done[cls] = 1;
      }
      if (p->relax_alpha != 0.0) {                                       /* :457-469, :617-629 */
        orc_weight_rtpp(k, p->relax_alpha, tr, parm, wrlx);
      } else if (p->relax_alpha_spread != 0.0) {
        double xb[k], tmpinfl;
        for (int m = 0; m < k; ++m) xb[m] = g[m * sm + n * sv];
        orc_weight_rtps(k, p->relax_alpha_spread, tr, pc, xb, parm, wrlx, &tmpinfl);
      } else {
        memcpy(wrlx, tr, sizeof(double) * kk);
      }
      for (int m = 0; m < k; ++m) {                                      /* :472-477, :632-637 */
        for (int q = 0; q < k; ++q) CM(wrlx, k, q, m) = (CM(wrlx, k, q, m) + tm[q]) * beta;
        CM(wrlx, k, m, m) += (1.0 - beta);
      }
      for (int m = 0; m < k; ++m) {                                      /* :480-486, :640-646 */
        double t = g[k * sm + n * sv];
        for (int q = 0; q < k; ++q) t = t + g[q * sm + n * sv] * CM(wrlx, k, q, m);
        a[m * sm + n * sv] = t;
      }
      if (p->det_run) {                                                  /* :489-497, :649-657 */
        double t = 0.0;
        for (int q = 0; q < k; ++q) t = t + g[q * sm + n * sv] * tmd[q];
        a[(k + 1) * sm + n * sv] = g[(k + 1) * sm + n * sv] + t * beta;
      }
      if (!is2d && p->q_sprd_max > 0.0 && n == p->iv_q_first) {          /* :500-513 (3-D only) */
        double q_mean = 0.0, q_sprd = 0.0, q_anal[k];
        for (int m = 0; m < k; ++m) q_mean += a[m * sm + n * sv];
        q_mean /= (double)k;
        for (int m = 0; m < k; ++m) {
          q_anal[m] = a[m * sm + n * sv] - q_mean;
          q_sprd += q_anal[m] * q_anal[m];
        }
        q_sprd = sqrt(q_sprd / (double)(k - 1)) / q_mean;
        if (q_sprd > p->q_sprd_max)
          for (int m = 0; m < k; ++m) a[m * sm + n * sv] = q_mean + q_anal[m] * p->q_sprd_max / q_sprd;
      }
    }
  }
  free(hdxf);
  free(rd);
  free(trans);
  free(trans_done);
  return worst;
}

/* common/common_sort.f90:341-369 / :404-432 -- selection semantics only: after the call the first K entries of x
 * index the K smallest (largest) keys.  The reference's pivot rules (sample_second_min_arg :224, median_of_three_arg
 * :168) only decide the ORDER inside the two halves, which its callers never rely on
 * (scale/letkf/letkf_tools.f90:1615-1628 copies the first K in whatever order they come). */
typedef struct { double key; int32_t id; } orc_kv;
static int orc_kv_cmp(const void *pa, const void *pb) {
  const orc_kv *a = (const orc_kv *)pa, *b = (const orc_kv *)pb;
  if (a->key < b->key) return -1;
  if (a->key > b->key) return 1;
  return (a->id > b->id) - (a->id < b->id);
}
void orc_select_arg(const double *a, int32_t *x, int n, int K, int desc) {
  (void)K;
  orc_kv *kv = (orc_kv *)malloc(sizeof(orc_kv) * (size_t)(n > 0 ? n : 1));
  for (int i = 0; i < n; ++i) {
    kv[i].key = desc ? -a[x[i]] : a[x[i]];
    kv[i].id = x[i];
  }
  qsort(kv, (size_t)n, sizeof(orc_kv), orc_kv_cmp);
  for (int i = 0; i < n; ++i) x[i] = kv[i].id;
  free(kv);
}

/* scale/letkf/letkf_obs.f90:1209-1227 */
static void orc_ij_obsgrd_ext(const orc_search_tables *t, int ic, double ri, double rj, int *ogi, int *ogj) {
  *ogi = (int)ceil((ri - t->i_org) * (double)t->ngrd_i[ic] / (double)t->nlon) + t->ngrdsch_i[ic];
  *ogj = (int)ceil((rj - t->j_org) * (double)t->ngrd_j[ic] / (double)t->nlat) + t->ngrdsch_j[ic];
}

/* scale/letkf/letkf_tools.f90:1765-1788 */
static void orc_obs_local_range(const orc_search_tables *t, int ic, double ri, double rj, int *imin, int *imax,
                                int *jmin, int *jmax) {
  const double dzi = t->hori_loc[ic] * ORC_DIST_ZERO_FAC / t->dx;
  const double dzj = t->hori_loc[ic] * ORC_DIST_ZERO_FAC / t->dy;
  orc_ij_obsgrd_ext(t, ic, ri - dzi, rj - dzj, imin, jmin);
  orc_ij_obsgrd_ext(t, ic, ri + dzi, rj + dzj, imax, jmax);
}

static inline int32_t orc_ac(const orc_search_tables *t, int ic, int i, int j) {   /* obsgrd(ic)%ac_ext(i, j) */
  return t->ac_ext[t->ac_off[ic] + i + (int64_t)(t->ngrdext_i[ic] + 1) * (j - 1)];
}

static int obs_local_impl(const orc_search_tables *t, double ri, double rj, double rlev, double rz, int cap,
                          int32_t *idx_out, double *rdiag_out, double *rloc_out, double *dist_out, int *tied) {
  int nobsl = 0;
  if (tied) *tied = 0;
  for (int ig = 0; ig < t->ngroup; ++ig) {                       /* do ic = 1, nctype (masters only) :1426-1432 */
    const int gs = t->group_start[ig], ge = t->group_start[ig + 1];
    const int icm = t->group_member[gs];                         /* master ctype :1434-1436 */
    const int nmax = t->max_nobs[icm];
    /* candidates of the whole group inside the cut-off rectangles, in the reference's order:
     * member ctype, mesh row j, table row (obs_choose_ext, letkf_obs.f90:1262-1285) */
    int ncand = 0, ccap = 256;
    int32_t *cand = (int32_t *)malloc(sizeof(int32_t) * (size_t)ccap);
    double *cr = (double *)malloc(sizeof(double) * (size_t)ccap * 3);   /* rloc, rdiag, dist */
    for (int m = gs; m < ge; ++m) {
      const int ic = t->group_member[m];
      int imin, imax, jmin, jmax;
      orc_obs_local_range(t, ic, ri, rj, &imin, &imax, &jmin, &jmax);
      if (imin > imax || jmin > jmax) continue;
      for (int j = jmin; j <= jmax; ++j)
        for (int32_t row = orc_ac(t, ic, imin - 1, j); row < orc_ac(t, ic, imax, j); ++row) {
          double nd, nr;
          const double rl = orc_obs_local_cal(ri, rj, rlev, rz, t->varloc[ic], t->vmode[ic], t->hori_loc[ic],
                                              t->vert_loc[ic], t->rain_base, t->ob_ri[row], t->ob_rj[row],
                                              t->ob_lev[row], t->ob_dat[row], t->ob_err[row], t->dx, t->dy, &nd, &nr);
          if (rl == 0.0) continue;                               /* :1460 / :1584 / :1684 */
          if (ncand == ccap) {
            ccap *= 2;
            cand = (int32_t *)realloc(cand, sizeof(int32_t) * (size_t)ccap);
            cr = (double *)realloc(cr, sizeof(double) * (size_t)ccap * 3);
          }
          cand[ncand] = row;
          cr[3 * ncand] = rl;
          cr[3 * ncand + 1] = nr;
          cr[3 * ncand + 2] = nd;
          ++ncand;
        }
    }
    int nsel = ncand;
    int32_t *order = (int32_t *)malloc(sizeof(int32_t) * (size_t)(ncand > 0 ? ncand : 1));
    for (int i = 0; i < ncand; ++i) order[i] = i;
    if (nmax > 0 && ncand > nmax) {
      /* :1614-1617 (criterion 1: the N nearest inside the cut-off -- the incremental rectangle search of
       * :1527-1602 only bounds the work, any obs within normalised distance q*search_incr/hori_loc lies inside the
       * q-th rectangle, so the selected set is the same), :1694-1704 (criterion 2: largest rloc, 3: smallest rdiag) */
      double *key = (double *)malloc(sizeof(double) * (size_t)ncand);
      for (int i = 0; i < ncand; ++i)
        key[i] = (t->criterion == 1) ? cr[3 * i + 2] : (t->criterion == 2) ? cr[3 * i] : cr[3 * i + 1];
      orc_select_arg(key, order, ncand, nmax, t->criterion == 2);
      /* the last key selected equals the first one rejected: WHICH of them the reference keeps is up to its unstable
       * quick-select (common/common_sort.f90:341-369) -- the caller may want to know that this point's list is one of several */
      if (tied && key[order[nmax - 1]] == key[order[nmax]]) *tied = 1;
      free(key);
      nsel = nmax;
    }
    for (int s = 0; s < nsel; ++s) {
      const int i = order[s];
      if (nobsl >= cap) {
        free(cand); free(cr); free(order);
        return -1;
      }
      idx_out[nobsl] = cand[i];
      rloc_out[nobsl] = cr[3 * i];
      rdiag_out[nobsl] = cr[3 * i + 1];
      if (dist_out) dist_out[nobsl] = cr[3 * i + 2];
      ++nobsl;
    }
    free(cand);
    free(cr);
    free(order);
  }
  return nobsl;
}

int orc_obs_local(const orc_search_tables *t, double ri, double rj, double rlev, double rz, int cap,
                  int32_t *idx_out, double *rdiag_out, double *rloc_out, double *dist_out) {
  return obs_local_impl(t, ri, rj, rlev, rz, cap, idx_out, rdiag_out, rloc_out, dist_out, NULL);
}

int orc_obs_local_tied(const orc_search_tables *t, double ri, double rj, double rlev, double rz, int cap,
                       int32_t *idx_out, double *rdiag_out, double *rloc_out, double *dist_out, int *tied_out) {
  return obs_local_impl(t, ri, rj, rlev, rz, cap, idx_out, rdiag_out, rloc_out, dist_out, tied_out);
}

/* scale/common/common_scale.f90:1181-1224 and :1229-1280 */
void orc_state_trans(const orc_state_consts *c, int nlev, int nlon, int nlat, int nv3d, double *v, int inverse) {
  const int64_t nxy = (int64_t)nlon * nlat, st = (int64_t)nlev * nxy;
  if (inverse) {                                             /* :1243-1250 */
    for (int n = c->iv_q; n < nv3d; ++n) {
      const int clamp = (n == c->iv_q) ? c->positive_definite_q : c->positive_definite_qhyd;
      if (clamp)
        for (int64_t e = 0; e < st; ++e) v[n * st + e] = fmax(v[n * st + e], 0.0);
    }
  }
#pragma omp parallel for schedule(static)
  for (int64_t e = 0; e < st; ++e) {
    double *p = v + e;
    double qdry = 1.0, cvtot = 0.0;
    for (int n = c->iv_q; n < nv3d; ++n) {
      qdry = qdry - p[n * st];
      cvtot = cvtot + p[n * st] * c->tracer_cv[n - c->iv_q];
    }
    cvtot = c->cvdry * qdry + cvtot;
    const double rtot = c->rdry * qdry + c->rvap * p[c->iv_q * st];
    if (!inverse) {
      const double cpovcv = (cvtot + rtot) / cvtot;
      const double rho = p[c->iv_rho * st];
      const double pres = c->pre00 * pow(p[c->iv_rhot * st] * rtot / c->pre00, cpovcv);
      const double temp = pres / (rho * rtot);
      const double u = p[c->iv_rhou * st] / rho, vv = p[c->iv_rhov * st] / rho, w = p[c->iv_rhow * st] / rho;
      p[c->iv_u * st] = u;
      p[c->iv_v * st] = vv;
      p[c->iv_w * st] = w;
      p[c->iv_t * st] = temp;
      p[c->iv_p * st] = pres;
    } else {
      const double cvovcp = cvtot / (cvtot + rtot);
      const double pr = p[c->iv_p * st];
      const double rho = pr / (rtot * p[c->iv_t * st]);
      const double rhot = c->pre00 / rtot * pow(pr / c->pre00, cvovcp);
      const double ru = p[c->iv_u * st] * rho, rv = p[c->iv_v * st] * rho, rw = p[c->iv_w * st] * rho;
      p[c->iv_rhot * st] = rhot;
      p[c->iv_rhow * st] = rw;
      p[c->iv_rhov * st] = rv;
      p[c->iv_rhou * st] = ru;
      p[c->iv_rho * st] = rho;
    }
  }
}

/* scale/common/common_mpi_scale.f90:1428-1455 (grd_to_buf), :1460-1480 (buf_to_grd) */
void orc_member_points(int dir, int nlev, int nlon, int nlat, int nv3d, int np, int rank, int m, double *v3dg,
                       double *x, int64_t nij1, int64_t sp, int64_t sm, int64_t sv) {
  const int64_t nxy = (int64_t)nlon * nlat;
  for (int n = 0; n < nv3d; ++n)
    for (int k = 0; k < nlev; ++k)
      for (int64_t i = 0; i < nij1; ++i) {
        const int64_t j = rank + (int64_t)np * i;            /* j = m-1 + np*(i-1) */
        const int ilon = (int)(j % nlon);
        const int ilat = (int)((j - ilon) / nlon);
        double *f = &v3dg[k + (int64_t)nlev * (ilon + (int64_t)nlon * ilat) + (int64_t)n * nlev * nxy];
        double *b = &x[(i + nij1 * k) * sp + (int64_t)m * sm + (int64_t)n * sv];
        if (dir == 0) *b = *f;
        else *f = *b;
      }
}

/* scale/common/common_scale.f90:1570-1607 */
void orc_ens_spread(int k, int nv, int64_t npts, const double *x, int64_t sp, int64_t sm, int64_t sv, double *sprd) {
  for (int v = 0; v < nv; ++v)
    for (int64_t pt = 0; pt < npts; ++pt) {
      const double *b = x + pt * sp + v * sv;
      const double mean = b[k * sm];
      double s = (b[0] - mean) * (b[0] - mean);
      for (int m = 1; m < k; ++m) s = s + (b[m * sm] - mean) * (b[m * sm] - mean);
      sprd[pt + npts * v] = sqrt(s / (double)(k - 1));
    }
}


/* ------------------------------------------------------------------------------------------------
 * Row f2: set_letkf_obs
 * ---------------------------------------------------------------------------------------------- */
enum {
  ORC_ID_RAIN = 19999, ORC_ID_RADAR_REF = 4001, ORC_ID_RADAR_REF_ZERO = 4004, ORC_ID_RADAR_VR = 4002,
  ORC_ID_RADAR_PRH = 4003, ORC_ID_TCLON = 99991, ORC_ID_TCLAT = 99992, ORC_ID_TCMIP = 99993, ORC_ID_H08IR = 8800,   /* common_obs_scale.f90:48-72 */
  ORC_QC_GOOD = 0, ORC_QC_GROSS = 5, ORC_QC_REF_MEM = 12, ORC_QC_OBS_BAD = 50, ORC_QC_OTYPE = 90   /* :139-151 */
};
static const double ORC_UNDEF = -9.99e33;   /* common/common.f90:38 */

void orc_obs_departure(const orc_qc_params *p, int64_t nobs, const int32_t *elm, const double *dat, const double *err,
                       double *ensval, int64_t kld, double *val, int32_t *qc) {
  const int K = p->member;
  for (int64_t n = 0; n < nobs; ++n) {
    if (qc[n] > 0) continue;                                               /* letkf_obs.f90:362 */
    double *e = ensval + n * kld;
    const int el = elm[n];
    if (el == ORC_ID_RADAR_REF || el == ORC_ID_RADAR_REF_ZERO) {           /* :372-411 */
      if (!p->use_radar_ref) { qc[n] = ORC_QC_OTYPE; continue; }
      if (dat[n] == ORC_UNDEF) { qc[n] = ORC_QC_OBS_BAD; continue; }
      int mem_ref = 0;
      for (int i = 0; i < K; ++i)
        if (e[i] > p->radar_ref_thres_dbz + 1.0e-6) ++mem_ref;
      if (dat[n] > p->radar_ref_thres_dbz + 1.0e-6) {
        if (mem_ref < p->min_radar_ref_member_obsref) { qc[n] = ORC_QC_REF_MEM; continue; }
      } else {
        if (mem_ref < p->min_radar_ref_member) { qc[n] = ORC_QC_REF_MEM; continue; }
      }
    }
    if (el == ORC_ID_RADAR_VR && !p->use_radar_vr) { qc[n] = ORC_QC_OTYPE; continue; }   /* :413-418 */
    int mem_cld = 0;
    if (p->h08 && el == ORC_ID_H08IR) {                                     /* #ifdef H08, :432-469 */
      if (dat[n] == ORC_UNDEF) { qc[n] = ORC_QC_OBS_BAD; continue; }
      if (p->h08_lev[n] < p->h08_limit_lev) { qc[n] = ORC_QC_OBS_BAD; continue; }
      for (int i = 0; i < K; ++i)
        if (e[i] < 0.0) {
          mem_cld = mem_cld + 1;
          e[i] = e[i] * (-1.0);
        }
    }
    double v = e[0];                                                        /* :475-479 */
    for (int i = 1; i < K; ++i) v = v + e[i];
    v = v / (double)K;
    if (p->h08 && p->h08_val2)                                              /* #ifdef H08, :480-487 */
      p->h08_val2[n] = (fabs(v - p->h08_val2[n]) + fabs(dat[n] - p->h08_val2[n])) * 0.5;
    for (int i = 0; i < K; ++i) e[i] = e[i] - v;                            /* :488-490 */
    v = dat[n] - v;                                                         /* :491 */
    val[n] = v;
    if (p->det_run) e[K] = dat[n] - e[K];                                   /* :492-494 */
    double ge;                                                              /* :504-561 */
    switch (el) {
      case ORC_ID_RAIN: ge = p->gross_error_rain; break;
      case ORC_ID_RADAR_REF:
      case ORC_ID_RADAR_REF_ZERO: ge = p->gross_error_radar_ref; break;
      case ORC_ID_RADAR_VR: ge = p->gross_error_radar_vr; break;
      case ORC_ID_RADAR_PRH: ge = p->gross_error_radar_prh; break;
      case ORC_ID_TCLON: ge = p->gross_error_tcx; break;
      case ORC_ID_TCLAT: ge = p->gross_error_tcy; break;
      case ORC_ID_TCMIP: ge = p->gross_error_tcp; break;
      default: ge = p->gross_error;
    }
    if (p->h08 && el == ORC_ID_H08IR) {                                     /* case (id_H08IR_obs), :520-541.  (The case itself
                                                                               is outside the #ifdef; in a default build mem_ref
                                                                               is never assigned for such a row -- undefined --
                                                                               and no default-build obs operator produces one:
                                                                               h08 = 0 treats it as an ordinary row.) */
      if (mem_cld < p->h08_min_cld_member) {
        if (fabs(v) > 1.0 * err[n]) qc[n] = ORC_QC_GROSS;
      } else {
        if (fabs(v) > p->gross_error_h08 * err[n]) qc[n] = ORC_QC_GROSS;
      }
      if (dat[n] < p->h08_bt_min) qc[n] = ORC_QC_GROSS;
    } else if (fabs(v) > ge * err[n]) qc[n] = ORC_QC_GROSS;
  }
}

/* ij_obsgrd, letkf_obs.f90:1186-1203 (rij_g2l: common_scale.f90:1683-1697).  The reference scales rj by ngrd_i,
 * not ngrd_j (:1200); restated as written.  Returns 1-based mesh indices after the clamps of :768-771. */
static void orc_ij_obsgrd(const orc_mesh *m, int ic, double ri, double rj, int *ogi, int *ogj) {
  const double ril = ri - (double)(m->rank_i * m->nlon);
  const double rjl = rj - (double)(m->rank_j * m->nlat);
  int i = (int)ceil((ril - (double)m->ihalo - 0.5) * (double)m->ngrd_i[ic] / (double)m->nlon);
  int j = (int)ceil((rjl - (double)m->jhalo - 0.5) * (double)(m->fix_ij_obsgrd ? m->ngrd_j[ic] : m->ngrd_i[ic]) / (double)m->nlat);
  if (i < 1) i = 1;
  if (i > m->ngrd_i[ic]) i = m->ngrd_i[ic];
  if (j < 1) j = 1;
  if (j > m->ngrd_j[ic]) j = m->ngrd_j[ic];
  *ogi = i;
  *ogj = j;
}

int64_t orc_obs_mesh_sort(const orc_mesh *m, int64_t nobs, const int32_t *ctype, const double *ri, const double *rj,
                          const int32_t *qc, int32_t *n_cell, int32_t *key) {
  int64_t ncell = 0;
  int64_t *coff = (int64_t *)malloc(sizeof(int64_t) * (size_t)(m->nctype + 1));
  for (int ic = 0; ic < m->nctype; ++ic) {
    coff[ic] = ncell;
    ncell += (int64_t)m->ngrd_i[ic] * m->ngrd_j[ic];
  }
  coff[m->nctype] = ncell;
  for (int64_t c = 0; c < ncell; ++c) n_cell[c] = 0;
  /* first scan (:762-781) */
  for (int64_t n = 0; n < nobs; ++n) {
    if (qc[n] != ORC_QC_GOOD) continue;
    int i, j;
    const int ic = ctype[n];
    orc_ij_obsgrd(m, ic, ri[n], rj[n], &i, &j);
    n_cell[coff[ic] + (int64_t)(j - 1) * m->ngrd_i[ic] + (i - 1)] += 1;
  }
  /* accumulated numbers (:786-800): cells in (ctype, j, i) order */
  int64_t *next = (int64_t *)malloc(sizeof(int64_t) * (size_t)(ncell > 0 ? ncell : 1));
  int64_t acc = 0;
  for (int64_t c = 0; c < ncell; ++c) {
    next[c] = acc;
    acc += n_cell[c];
  }
  /* second scan (:806-822) */
  for (int64_t n = 0; n < nobs; ++n) {
    if (qc[n] != ORC_QC_GOOD) continue;
    int i, j;
    const int ic = ctype[n];
    orc_ij_obsgrd(m, ic, ri[n], rj[n], &i, &j);
    const int64_t c = coff[ic] + (int64_t)(j - 1) * m->ngrd_i[ic] + (i - 1);
    key[next[c]++] = (int32_t)n;
  }
  free(next);
  free(coff);
  return acc;
}

int64_t orc_obs_halo_plan(const orc_halo_layout *l, const int32_t *n_all, int32_t *ac_ext, int32_t *src_row,
                          int64_t cap) {
  const int nc = l->nctype, np = l->nprocs;
  int64_t ncell = 0, nacx = 0;
  int64_t *coff = (int64_t *)malloc(sizeof(int64_t) * (size_t)(nc + 1));
  int64_t *xoff = (int64_t *)malloc(sizeof(int64_t) * (size_t)(nc + 1));
  for (int ic = 0; ic < nc; ++ic) {
    coff[ic] = ncell;
    xoff[ic] = nacx;
    ncell += (int64_t)l->ngrd_i[ic] * l->ngrd_j[ic];
    nacx += (int64_t)(l->ngrd_i[ic] + 2 * l->ngrdsch_i[ic] + 1) * (l->ngrd_j[ic] + 2 * l->ngrdsch_j[ic]);
  }
  /* every rank's accumulated counts ac(i,j,ip) (exclusive start of each cell), and the displacements dspr (:979-984) */
  int64_t *start = (int64_t *)malloc(sizeof(int64_t) * (size_t)(np * (ncell > 0 ? ncell : 1)));
  int64_t *dspr = (int64_t *)malloc(sizeof(int64_t) * (size_t)(np + 1));
  dspr[0] = 0;
  for (int ip = 0; ip < np; ++ip) {
    int64_t acc = 0;
    for (int64_t c = 0; c < ncell; ++c) {
      start[ip * ncell + c] = acc;
      acc += n_all[ip * ncell + c];
    }
    dspr[ip + 1] = dspr[ip] + acc;
  }
  const int myp_i = l->myrank % l->prc_num_x, myp_j = l->myrank / l->prc_num_x;
  int64_t acx = 0;       /* running ac_ext over ctypes (:946-956) */
  int64_t rc = 0;
  for (int ic = 0; ic < nc && rc >= 0; ++ic) {
    const int gi = l->ngrd_i[ic], gj = l->ngrd_j[ic], si = l->ngrdsch_i[ic], sj = l->ngrdsch_j[ic];
    const int ei = gi + 2 * si, ej = gj + 2 * sj;
    int32_t *nx = (int32_t *)calloc((size_t)ei * ej, sizeof(int32_t));
    int64_t *sx = (int64_t *)malloc(sizeof(int64_t) * (size_t)ei * ej);   /* source row of the first obs of the cell */
    const int imin1 = myp_i * gi + 1 - si, imax1 = (myp_i + 1) * gi + si;  /* :925-928 */
    const int jmin1 = myp_j * gj + 1 - sj, jmax1 = (myp_j + 1) * gj + sj;
    for (int ip = 0; ip < np; ++ip) {
      const int ip_i = ip % l->prc_num_x, ip_j = ip / l->prc_num_x;
      int imin2 = imin1 - ip_i * gi, imax2 = imax1 - ip_i * gi;            /* :932-936 */
      int jmin2 = jmin1 - ip_j * gj, jmax2 = jmax1 - ip_j * gj;
      if (imin2 < 1) imin2 = 1;
      if (imax2 > gi) imax2 = gi;
      if (jmin2 < 1) jmin2 = 1;
      if (jmax2 > gj) jmax2 = gj;
      if (imin2 > imax2 || jmin2 > jmax2) continue;
      const int ishift = (ip_i - myp_i) * gi + si, jshift = (ip_j - myp_j) * gj + sj;   /* :938-940 */
      for (int j = jmin2; j <= jmax2; ++j)
        for (int i = imin2; i <= imax2; ++i) {
          const int64_t c = coff[ic] + (int64_t)(j - 1) * gi + (i - 1);
          const int64_t x = (int64_t)(j + jshift - 1) * ei + (i + ishift - 1);
          nx[x] = n_all[ip * ncell + c];
          sx[x] = dspr[ip] + start[ip * ncell + c];
        }
    }
    int32_t *ax = ac_ext + xoff[ic];                                       /* [ej][ei + 1] */
    for (int j = 0; j < ej; ++j) {
      ax[(int64_t)j * (ei + 1)] = (int32_t)acx;
      for (int i = 0; i < ei; ++i) {
        const int64_t x = (int64_t)j * ei + i;
        if (nx[x] > 0) {
          if (acx + nx[x] > cap) { rc = -1; break; }
          for (int t = 0; t < nx[x]; ++t) src_row[acx + t] = (int32_t)(sx[x] + t);   /* :1060-1080, cell by cell */
        }
        acx += nx[x];
        ax[(int64_t)j * (ei + 1) + i + 1] = (int32_t)acx;
      }
      if (rc < 0) break;
    }
    free(nx);
    free(sx);
  }
  free(start);
  free(dspr);
  free(coff);
  free(xoff);
  return rc < 0 ? -1 : acx;
}


/* ------------------------------------------------------------------------------------------------
 * Row f4
 * ---------------------------------------------------------------------------------------------- */
void orc_monit_dep(int nid, const int32_t *elem_uid, int64_t nn, const int32_t *elm, const double *dep,
                   const int32_t *qc, int32_t *nobs, double *bias, double *rmse) {
  for (int i = 0; i < nid; ++i) {
    nobs[i] = 0;
    bias[i] = 0.0;
    rmse[i] = 0.0;
  }
  for (int64_t n = 0; n < nn; ++n) {                                   /* common_obs_scale.f90:1867-1882 */
    if (qc[n] != ORC_QC_GOOD) continue;
    int ielm = elm[n];
    if (ielm == 3074) ielm = 3073;                                     /* Tv as T */
    if (ielm == ORC_ID_RADAR_REF_ZERO) ielm = ORC_ID_RADAR_REF;        /* RE0 as REF */
    int i = -1;
    for (int u = 0; u < nid; ++u)
      if (elem_uid[u] == ielm) i = u;
    if (i < 0) continue;
    nobs[i] += 1;
    bias[i] = bias[i] + dep[n];
    rmse[i] = rmse[i] + dep[n] * dep[n];
  }
  for (int i = 0; i < nid; ++i) {                                      /* :1884-1892 */
    if (nobs[i] == 0) {
      bias[i] = ORC_UNDEF;
      rmse[i] = ORC_UNDEF;
    } else {
      bias[i] = bias[i] / (double)nobs[i];
      rmse[i] = sqrt(rmse[i] / (double)nobs[i]);
    }
  }
}

void orc_additive_inflation(int k, int nv, int64_t npts, int64_t nij1, double *anal, const double *add, int64_t sp,
                            int64_t sm, int64_t sv, double infl_add, const double *weight, const double *qmean,
                            int64_t q_sp, int64_t q_sv, int iv_q_first, int iv_q_last, const int32_t *ishuf) {
  for (int v = 0; v < nv; ++v)                                         /* letkf_tools.f90:884-913 */
    for (int m = 0; m < k; ++m) {
      const int ms = ishuf ? ishuf[m] : m;
      for (int64_t p = 0; p < npts; ++p) {
        double x = add[p * sp + ms * sm + v * sv] * infl_add;
        x = x * (weight ? weight[p % nij1] : 1.0);
        if (qmean && v >= iv_q_first && v <= iv_q_last) x = x * qmean[p * q_sp + v * q_sv];
        anal[p * sp + m * sm + v * sv] = anal[p * sp + m * sm + v * sv] + x;
      }
    }
}

void orc_addinfl_weight(int64_t nij1, const double *rig, const double *rjg, int64_t nob, const double *ob_ri,
                        const double *ob_rj, double dx, double dy, double hori_loc, double *w) {
  const double cut2 = (double)13.33333333f;                            /* dist_zero_fac_square, letkf_obs.f90:28 */
  for (int64_t ij = 0; ij < nij1; ++ij) {                              /* letkf_tools.f90:818-836 */
    double ref_min_dist = 1.0e33;
    for (int64_t o = 0; o < nob; ++o) {
      const double rdx = (rig[ij] - ob_ri[o]) * dx, rdy = (rjg[ij] - ob_rj[o]) * dy;
      const double rdxy = rdx * rdx + rdy * rdy;
      if (rdxy < ref_min_dist) ref_min_dist = rdxy;
    }
    ref_min_dist = ref_min_dist / (hori_loc * hori_loc);
    w[ij] = ref_min_dist <= cut2 ? exp(-0.5 * ref_min_dist) : 0.0;
  }
}
