/*
 * TEST INFRASTRUCTURE ONLY -- CPU restatement of the arithmetic on the GCN
 * propagation path.  Nothing under gcn_amd/ may link, load or call this file;
 * it exists so that tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg have an independent checker for the HIP kernels.
 *
 * The arithmetic itself lives in TensorFlow 1.x (setup.py:13 pins
 * tensorflow>=1.15.2,<2.0), which is not vendored in the reference tree and
 * not installable here, so the TF kernels are restated from their published
 * semantics and anchored on the reference's call sites:
 *   tf.sparse_tensor_dense_matmul  gcn/layers.py:33   -> orc_spmm_coo_f32
 *   its gradient (adjoint_a=True)  gcn/models.py:58   -> orc_spmm_coo_adj_f32
 *   tf.sparse_retain + 1/keep      gcn/layers.py:21-27-> orc_sparse_dropout_f32
 *   tf.matmul                      gcn/layers.py:35   -> orc_gemm_f32
 *   softmax_cross_entropy + mask   gcn/metrics.py:4-10-> orc_masked_softmax_xent_f32
 *   argmax equality + mask         gcn/metrics.py:13-20 -> orc_masked_accuracy_f32
 *   tf.train.AdamOptimizer         gcn/models.py:142  -> orc_adam_step_f32
 *   D^-1/2 A D^-1/2                gcn/utils.py:122-129 -> orc_normalize_coo_f64
 * PARITY UNPINNED for the TF-side functions: the reference holds no tests or
 * golden vectors for them; they are cross-checked against scipy.sparse and
 * torch CPU autograd in tests/.  The preprocessing restatement IS pinned by
 * golden vectors produced by the reference's own gcn/utils.py.
 *
 * Plain C99; OpenMP only in the functions whose name ends in _omp.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

int orc_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* ---- SpMM, TF CPU functor order: walk the nonzeros in the order given and do
 * out.row(row[k]) += val[k] * b.row(col[k]) in fp32 (gcn/layers.py:33). ---- */
void orc_spmm_coo_f32(int64_t n_rows, int64_t nnz, const int32_t *row,
                      const int32_t *col, const float *val, const float *b,
                      int64_t ldb, float *out, int64_t ldo, int32_t width) {
  for (int64_t i = 0; i < n_rows; ++i)
    memset(out + i * ldo, 0, sizeof(float) * (size_t)width);
  for (int64_t k = 0; k < nnz; ++k) {
    const float v = val[k];
    const float *src = b + (int64_t)col[k] * ldb;
    float *dst = out + (int64_t)row[k] * ldo;
    for (int32_t j = 0; j < width; ++j) dst[j] += v * src[j];
  }
}

/* Gradient of the above w.r.t. the dense operand: the same functor with the
 * sparse operand adjointed, out.row(col[k]) += val[k] * g.row(row[k])
 * (created by optimizer.minimize, gcn/models.py:58). */
void orc_spmm_coo_adj_f32(int64_t n_cols, int64_t nnz, const int32_t *row,
                          const int32_t *col, const float *val, const float *g,
                          int64_t ldg, float *out, int64_t ldo, int32_t width) {
  for (int64_t i = 0; i < n_cols; ++i)
    memset(out + i * ldo, 0, sizeof(float) * (size_t)width);
  for (int64_t k = 0; k < nnz; ++k) {
    const float v = val[k];
    const float *src = g + (int64_t)row[k] * ldg;
    float *dst = out + (int64_t)col[k] * ldo;
    for (int32_t j = 0; j < width; ++j) dst[j] += v * src[j];
  }
}

/* CSR row-wise product, sequential over each row's nonzeros, fp32 multiply and
 * add (same sum as orc_spmm_coo_f32 when the COO is row-major sorted). */
void orc_spmm_csr_f32(int64_t n_rows, const int64_t *row_ptr,
                      const int32_t *col_idx, const float *val, const float *b,
                      int64_t ldb, float *out, int64_t ldo, int32_t width) {
  for (int64_t i = 0; i < n_rows; ++i) {
    float *dst = out + i * ldo;
    memset(dst, 0, sizeof(float) * (size_t)width);
    for (int64_t k = row_ptr[i]; k < row_ptr[i + 1]; ++k) {
      const float v = val[k];
      const float *src = b + (int64_t)col_idx[k] * ldb;
      for (int32_t j = 0; j < width; ++j) dst[j] += v * src[j];
    }
  }
}

/* The multi-core CPU number: rows distributed over OpenMP threads. */
void orc_spmm_csr_f32_omp(int64_t n_rows, const int64_t *row_ptr,
                          const int32_t *col_idx, const float *val,
                          const float *b, int64_t ldb, float *out, int64_t ldo,
                          int32_t width) {
#pragma omp parallel for schedule(dynamic, 256)
  for (int64_t i = 0; i < n_rows; ++i) {
    float *dst = out + i * ldo;
    memset(dst, 0, sizeof(float) * (size_t)width);
    for (int64_t k = row_ptr[i]; k < row_ptr[i + 1]; ++k) {
      const float v = val[k];
      const float *src = b + (int64_t)col_idx[k] * ldb;
      for (int32_t j = 0; j < width; ++j) dst[j] += v * src[j];
    }
  }
}

/* Same product with fp64 accumulation: the yardstick the fp32 tolerance is
 * measured against.  abs_out (nullable) receives sum |val*b|. */
void orc_spmm_csr_f64acc(int64_t n_rows, const int64_t *row_ptr,
                         const int32_t *col_idx, const float *val,
                         const float *b, int64_t ldb, double *out,
                         double *abs_out, int64_t ldo, int32_t width) {
#pragma omp parallel for schedule(dynamic, 256)
  for (int64_t i = 0; i < n_rows; ++i) {
    double *dst = out + i * ldo;
    double *adst = abs_out ? abs_out + i * ldo : 0;
    for (int32_t j = 0; j < width; ++j) dst[j] = 0.0;
    if (adst) for (int32_t j = 0; j < width; ++j) adst[j] = 0.0;
    for (int64_t k = row_ptr[i]; k < row_ptr[i + 1]; ++k) {
      const double v = val[k];
      const float *src = b + (int64_t)col_idx[k] * ldb;
      for (int32_t j = 0; j < width; ++j) {
        dst[j] += v * (double)src[j];
        if (adst) adst[j] += fabs(v * (double)src[j]);
      }
    }
  }
}

/* ---- COO -> CSR: stable order by (row, col); duplicates are kept, in input
 * order; fp64 values are rounded to fp32 (round-to-nearest-even) last, which is
 * what feeding an fp64 tuple to a tf.float32 sparse placeholder does
 * (gcn/train_normal.py:50, gcn/utils.py:144).  perm[slot] = input index. ---- */
typedef struct { int32_t r, c; int64_t k; } orc_ent;
static int orc_ent_cmp(const void *a, const void *b) {
  const orc_ent *x = (const orc_ent *)a, *y = (const orc_ent *)b;
  if (x->r != y->r) return x->r < y->r ? -1 : 1;
  if (x->c != y->c) return x->c < y->c ? -1 : 1;
  return x->k < y->k ? -1 : (x->k > y->k);
}
int orc_csr_from_coo(int64_t n_rows, int64_t n_cols, int64_t nnz,
                     const int32_t *row, const int32_t *col, const void *val,
                     int val_is_f64, int64_t *row_ptr, int32_t *col_idx,
                     float *val_out, int64_t *perm) {
  orc_ent *e = (orc_ent *)malloc(sizeof(orc_ent) * (size_t)(nnz ? nnz : 1));
  if (!e) return -1;
  for (int64_t k = 0; k < nnz; ++k) {
    if (row[k] < 0 || row[k] >= n_rows || col[k] < 0 || col[k] >= n_cols) {
      free(e);
      return -2;
    }
    e[k].r = row[k]; e[k].c = col[k]; e[k].k = k;
  }
  qsort(e, (size_t)nnz, sizeof(orc_ent), orc_ent_cmp);
  for (int64_t i = 0; i <= n_rows; ++i) row_ptr[i] = 0;
  for (int64_t s = 0; s < nnz; ++s) {
    row_ptr[e[s].r + 1]++;
    col_idx[s] = e[s].c;
    perm[s] = e[s].k;
    val_out[s] = val_is_f64 ? (float)((const double *)val)[e[s].k]
                            : ((const float *)val)[e[s].k];
  }
  for (int64_t i = 0; i < n_rows; ++i) row_ptr[i + 1] += row_ptr[i];
  free(e);
  return 0;
}

/* Structural transpose of a CSR (no arithmetic): output rows are the input
 * columns, entries inside an output row ordered by input row then input slot.
 * perm[out slot] = input slot. */
void orc_csr_transpose(int64_t n_rows, int64_t n_cols, const int64_t *row_ptr,
                       const int32_t *col_idx, const float *val,
                       int64_t *t_row_ptr, int32_t *t_col_idx, float *t_val,
                       int64_t *perm) {
  const int64_t nnz = row_ptr[n_rows];
  for (int64_t i = 0; i <= n_cols; ++i) t_row_ptr[i] = 0;
  for (int64_t k = 0; k < nnz; ++k) t_row_ptr[col_idx[k] + 1]++;
  for (int64_t i = 0; i < n_cols; ++i) t_row_ptr[i + 1] += t_row_ptr[i];
  int64_t *fill = (int64_t *)malloc(sizeof(int64_t) * (size_t)(n_cols ? n_cols : 1));
  memcpy(fill, t_row_ptr, sizeof(int64_t) * (size_t)n_cols);
  for (int64_t i = 0; i < n_rows; ++i)
    for (int64_t k = row_ptr[i]; k < row_ptr[i + 1]; ++k) {
      const int64_t s = fill[col_idx[k]]++;
      t_col_idx[s] = (int32_t)i;
      t_val[s] = val[k];
      perm[s] = k;
    }
  free(fill);
}

/* ---- sparse dropout on the values of a sparse tensor (gcn/layers.py:21-27):
 * entries whose mask is 0 are dropped, survivors are multiplied by 1/keep.
 * The mask is an input (floor(keep + U[0,1)) is drawn by the caller) so that
 * both sides of a parity test see the same one.  Dropped entries are written as
 * +0, which contributes nothing to any later product. ---- */
void orc_sparse_dropout_f32(int64_t nnz, const float *val, const uint8_t *mask,
                            float keep, float *out) {
  const float scale = 1.0f / keep;
  for (int64_t k = 0; k < nnz; ++k) out[k] = mask[k] ? val[k] * scale : 0.0f;
}

/* ---- dense product, fp32, k-ordered accumulation (gcn/layers.py:35) ---- */
void orc_gemm_f32(int trans_a, int trans_b, int64_t M, int64_t N, int64_t K,
                  const float *A, int64_t lda, const float *B, int64_t ldb,
                  float *C, int64_t ldc, float alpha, float beta) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < M; ++i)
    for (int64_t j = 0; j < N; ++j) {
      float acc = 0.0f;
      for (int64_t k = 0; k < K; ++k) {
        const float a = trans_a ? A[k * lda + i] : A[i * lda + k];
        const float b = trans_b ? B[j * ldb + k] : B[k * ldb + j];
        acc += a * b;
      }
      C[i * ldc + j] = beta == 0.0f ? alpha * acc : alpha * acc + beta * C[i * ldc + j];
    }
}

/* fp64 yardstick of the same product. */
void orc_gemm_f64acc(int trans_a, int trans_b, int64_t M, int64_t N, int64_t K,
                     const float *A, int64_t lda, const float *B, int64_t ldb,
                     double *C, int64_t ldc) {
#pragma omp parallel for schedule(static)
  for (int64_t i = 0; i < M; ++i)
    for (int64_t j = 0; j < N; ++j) {
      double acc = 0.0;
      for (int64_t k = 0; k < K; ++k) {
        const double a = trans_a ? A[k * lda + i] : A[i * lda + k];
        const double b = trans_b ? B[j * ldb + k] : B[k * ldb + j];
        acc += a * b;
      }
      C[i * ldc + j] = acc;
    }
}

/* ---- masked softmax cross-entropy (gcn/metrics.py:4-10):
 * loss = mean_i( CE_i * mask_i / mean(mask) ), mean over ALL n rows, so the
 * per-row weight is mask_i / sum(mask).  grad (nullable) receives dloss/dlogits
 * = (softmax - labels * sum(labels_i)) * weight  [labels one-hot => softmax - y].
 * Accumulated in fp64, returned rounded to fp32. ---- */
float orc_masked_softmax_xent_f32(int64_t n, int32_t c, const float *logits,
                                  int64_t ld, const float *labels, int64_t ldl,
                                  const uint8_t *mask, float *grad, int64_t ldg) {
  double msum = 0.0;
  for (int64_t i = 0; i < n; ++i) msum += mask[i] ? 1.0 : 0.0;
  double total = 0.0;
  for (int64_t i = 0; i < n; ++i) {
    const float *z = logits + i * ld;
    const float *y = labels + i * ldl;
    float zmax = z[0];
    for (int32_t j = 1; j < c; ++j) zmax = z[j] > zmax ? z[j] : zmax;
    double se = 0.0, ysum = 0.0;
    for (int32_t j = 0; j < c; ++j) { se += exp((double)(z[j] - zmax)); ysum += y[j]; }
    const double lse = log(se);
    const double w = mask[i] ? 1.0 / msum : 0.0;
    double ce = 0.0;
    for (int32_t j = 0; j < c; ++j) ce -= (double)y[j] * ((double)(z[j] - zmax) - lse);
    total += ce * w;
    if (grad)
      for (int32_t j = 0; j < c; ++j)
        grad[i * ldg + j] =
            (float)((exp((double)(z[j] - zmax)) / se * ysum - (double)y[j]) * w);
  }
  return (float)total;
}

/* ---- masked accuracy (gcn/metrics.py:13-20); argmax takes the first maximum,
 * as tf.argmax does. ---- */
float orc_masked_accuracy_f32(int64_t n, int32_t c, const float *logits,
                              int64_t ld, const float *labels, int64_t ldl,
                              const uint8_t *mask) {
  double msum = 0.0, hit = 0.0;
  for (int64_t i = 0; i < n; ++i) {
    if (!mask[i]) continue;
    msum += 1.0;
    int32_t az = 0, ay = 0;
    for (int32_t j = 1; j < c; ++j) {
      if (logits[i * ld + j] > logits[i * ld + az]) az = j;
      if (labels[i * ldl + j] > labels[i * ldl + ay]) ay = j;
    }
    hit += az == ay ? 1.0 : 0.0;
  }
  return msum > 0 ? (float)(hit / msum) : 0.0f;
}

/* ---- Adam as TF 1.x applies it (gcn/models.py:142; TF defaults b1=.9,
 * b2=.999, eps=1e-8):  lr_t = lr*sqrt(1-b2^t)/(1-b1^t);  m,v moving averages;
 * w -= lr_t * m / (sqrt(v) + eps).  t counts from 1. ---- */
void orc_adam_step_f32(int64_t n, float *w, const float *g, float *m, float *v,
                       float lr, float b1, float b2, float eps, int64_t t) {
  const double lr_t = (double)lr * sqrt(1.0 - pow((double)b2, (double)t)) /
                      (1.0 - pow((double)b1, (double)t));
  const float lrt = (float)lr_t;
  for (int64_t i = 0; i < n; ++i) {
    m[i] = m[i] + (g[i] - m[i]) * (1.0f - b1);
    v[i] = v[i] + (g[i] * g[i] - v[i]) * (1.0f - b2);
    w[i] = w[i] - lrt * m[i] / (sqrtf(v[i]) + eps);
  }
}

/* ---- D^-1/2 A D^-1/2 on a COO in fp64 (gcn/utils.py:122-129): the value at
 * (r,c) of the result is fl(fl(a[c,r] * d[r]) * d[c]); the reference forms
 * (A.D)^T.D, so entry k of the input, at (row[k], col[k]), lands at
 * (col[k], row[k]) with value (val[k]*d[col[k]])*d[row[k]].  d is supplied by
 * the caller (numpy's rowsum ** -0.5 with inf -> 0). ---- */
void orc_normalize_coo_f64(int64_t nnz, const int32_t *row, const int32_t *col,
                           const double *val, const double *d, int32_t *out_row,
                           int32_t *out_col, double *out_val) {
  for (int64_t k = 0; k < nnz; ++k) {
    const double t = val[k] * d[col[k]];
    out_row[k] = col[k];
    out_col[k] = row[k];
    out_val[k] = t * d[row[k]];
  }
}
