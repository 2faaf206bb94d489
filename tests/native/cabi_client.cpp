// A client of the C ABI that knows nothing about Python or PyTorch: device
// memory from hipMalloc, a stream of its own, the calling sequences of
// include/gcnx.h (workspace query -> call), results checked against plain host
// loops in double precision.  Built by gcn_amd/csrc/Makefile into
// gcn_amd/lib/cabi_client; tests/test_gpu_kernels.py runs it.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <vector>

#include "gcnx.h"

#define HIP_OK(x)                                                                 \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) {                                                       \
      std::fprintf(stderr, "%s:%d: %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      return 2;                                                                   \
    }                                                                             \
  } while (0)
#define GCNX_OK_(x)                                                               \
  do {                                                                            \
    int rc_ = (x);                                                                \
    if (rc_ != GCNX_OK) {                                                         \
      std::fprintf(stderr, "%s:%d: status %d: %s\n", __FILE__, __LINE__, rc_, gcnx_last_error()); \
      return 3;                                                                   \
    }                                                                             \
  } while (0)

template <typename T>
static T *dev_alloc(size_t n) {
  void *p = nullptr;
  if (hipMalloc(&p, (n ? n : 1) * sizeof(T)) != hipSuccess) return nullptr;
  return static_cast<T *>(p);
}

static uint64_t rng_state = 0x9e3779b97f4a7c15ull;
static uint32_t rnd() {
  rng_state ^= rng_state << 13;
  rng_state ^= rng_state >> 7;
  rng_state ^= rng_state << 17;
  return (uint32_t)(rng_state >> 32);
}
static double urand() { return (rnd() >> 8) * (1.0 / 16777216.0) * 2.0 - 1.0; }

int main() {
  const int64_t n_rows = 5000, n_cols = 4000, nnz = 60000;
  const int32_t width = 64;
  const int64_t M = 3000, N = 16, K = 256;
  hipStream_t stream;
  HIP_OK(hipStreamCreate(&stream));

  // ---- COO -> CSR -> SpMM -------------------------------------------------
  std::vector<int32_t> row(nnz), col(nnz);
  std::vector<double> val(nnz);
  for (int64_t k = 0; k < nnz; ++k) {
    row[k] = (int32_t)(rnd() % n_rows);
    col[k] = (int32_t)(rnd() % n_cols);
    val[k] = urand();
  }
  for (int64_t k = 0; k < 3000; ++k) row[k] = 7;   // one long row
  std::vector<float> b((size_t)n_cols * width);
  for (auto &v : b) v = (float)urand();

  int32_t *d_row = dev_alloc<int32_t>(nnz), *d_col = dev_alloc<int32_t>(nnz), *d_ci = dev_alloc<int32_t>(nnz);
  double *d_val = dev_alloc<double>(nnz);
  int64_t *d_rp = dev_alloc<int64_t>(n_rows + 1);
  float *d_v = dev_alloc<float>(nnz), *d_b = dev_alloc<float>(b.size()), *d_c = dev_alloc<float>((size_t)n_rows * width);
  if (!d_row || !d_col || !d_ci || !d_val || !d_rp || !d_v || !d_b || !d_c) return 2;
  HIP_OK(hipMemcpy(d_row, row.data(), nnz * 4, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(d_col, col.data(), nnz * 4, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(d_val, val.data(), nnz * 8, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(d_b, b.data(), b.size() * 4, hipMemcpyHostToDevice));

  size_t ws_bytes = gcnx_csr_from_coo_workspace_bytes(n_rows, nnz);
  void *ws = dev_alloc<char>(ws_bytes);
  GCNX_OK_(gcnx_csr_from_coo(n_rows, n_cols, nnz, d_row, d_col, 1, d_val, 1, d_rp, d_ci, d_v, nullptr, ws,
                             ws_bytes, stream));
  gcnx_csr a = {n_rows, n_cols, nnz, d_rp, d_ci, d_v, 0, nullptr};
  int32_t *d_chunk = dev_alloc<int32_t>((size_t)gcnx_spmm_num_chunks(nnz) + 1);
  GCNX_OK_(gcnx_spmm_chunk_table(&a, d_chunk, stream));
  size_t sp_bytes = gcnx_spmm_workspace_bytes(&a, width);
  void *sp_ws = dev_alloc<char>(sp_bytes);
  GCNX_OK_(gcnx_spmm_csr_f32(&a, d_chunk, d_b, width, d_c, width, width, 1.0f, 0.0f, nullptr, nullptr, 1.0f, 1,
                             sp_ws, sp_bytes, stream));
  std::vector<float> c((size_t)n_rows * width);
  HIP_OK(hipStreamSynchronize(stream));
  HIP_OK(hipMemcpy(c.data(), d_c, c.size() * 4, hipMemcpyDeviceToHost));

  std::vector<double> ref((size_t)n_rows * width, 0.0), mag((size_t)n_rows * width, 0.0);
  for (int64_t k = 0; k < nnz; ++k) {
    const double v = (double)(float)val[k];
    for (int j = 0; j < width; ++j) {
      const double t = v * (double)b[(size_t)col[k] * width + j];
      ref[(size_t)row[k] * width + j] += t;
      mag[(size_t)row[k] * width + j] += std::fabs(t);
    }
  }
  double worst = 0.0;
  for (size_t i = 0; i < ref.size(); ++i) {
    const double want = ref[i] > 0.0 ? ref[i] : 0.0;   // fuse_relu = 1
    const double err = std::fabs((double)c[i] - want) / (mag[i] > 1e-30 ? mag[i] : 1e-30);
    if (err > worst) worst = err;
  }
  std::printf("spmm relu: worst error %.3g of the summed magnitudes\n", worst);
  if (!(worst <= 1e-5)) return 1;

  // ---- a wrong call comes back as a status, not as an abort ----------------
  if (gcnx_spmm_csr_f32(&a, d_chunk, d_b, width, d_c, width, width, 1.0f, 0.0f, nullptr, nullptr, 1.0f, 0,
                        sp_ws, 16, stream) != GCNX_ERR_WORKSPACE) return 1;

  // ---- cache hint: column degrees -> threshold -> keep bits; a 256-column product must not change ----
  {
    const int32_t ww = 256;
    int32_t *d_deg = dev_alloc<int32_t>((size_t)n_cols);
    uint64_t *d_keep = dev_alloc<uint64_t>((size_t)((nnz + 63) / 64));
    const size_t th_bytes = gcnx_csr_degree_threshold_workspace_bytes();
    void *th_ws = dev_alloc<char>(th_bytes);
    std::vector<float> bw((size_t)n_cols * ww);
    for (auto &v : bw) v = (float)urand();
    float *d_bw = dev_alloc<float>(bw.size());
    float *d_c1 = dev_alloc<float>((size_t)n_rows * ww), *d_c2 = dev_alloc<float>((size_t)n_rows * ww);
    if (!d_deg || !d_keep || !th_ws || !d_bw || !d_c1 || !d_c2) return 2;
    HIP_OK(hipMemcpy(d_bw, bw.data(), bw.size() * 4, hipMemcpyHostToDevice));
    GCNX_OK_(gcnx_csr_column_degrees(&a, d_deg, stream));
    int32_t min_degree = 0;
    GCNX_OK_(gcnx_csr_degree_threshold(d_deg, n_cols, n_cols / 8, &min_degree, th_ws, th_bytes, stream));
    GCNX_OK_(gcnx_csr_gather_keep(&a, d_deg, min_degree, d_keep, stream));
    std::vector<int32_t> deg((size_t)n_cols);
    HIP_OK(hipStreamSynchronize(stream));
    HIP_OK(hipMemcpy(deg.data(), d_deg, deg.size() * 4, hipMemcpyDeviceToHost));
    std::vector<int32_t> want_deg((size_t)n_cols, 0);
    for (int64_t k = 0; k < nnz; ++k) ++want_deg[col[k]];
    int64_t kept_cols = 0;
    for (int64_t c2 = 0; c2 < n_cols; ++c2) {
      if (deg[c2] != want_deg[c2]) return 1;
      kept_cols += deg[c2] >= min_degree;
    }
    if (min_degree < 2 || kept_cols > n_cols / 8) return 1;
    gcnx_csr hinted = a;
    hinted.spmm_chunk = 64;                 // the chunk kernel (the small-matrix kernel reads no hint)
    hinted.gather_keep = d_keep;
    gcnx_csr plain = hinted;
    plain.gather_keep = nullptr;
    const size_t hb = gcnx_spmm_workspace_bytes(&hinted, ww);
    void *h_ws = dev_alloc<char>(hb);
    if (!h_ws) return 2;
    GCNX_OK_(gcnx_spmm_csr_f32(&plain, d_chunk, d_bw, ww, d_c1, ww, ww, 1.0f, 0.0f, nullptr, nullptr, 1.0f, 1, h_ws, hb,
                               stream));
    GCNX_OK_(gcnx_spmm_csr_f32(&hinted, d_chunk, d_bw, ww, d_c2, ww, ww, 1.0f, 0.0f, nullptr, nullptr, 1.0f, 1, h_ws, hb,
                               stream));
    std::vector<float> c1v((size_t)n_rows * ww), c2v((size_t)n_rows * ww);
    HIP_OK(hipStreamSynchronize(stream));
    HIP_OK(hipMemcpy(c1v.data(), d_c1, c1v.size() * 4, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(c2v.data(), d_c2, c2v.size() * 4, hipMemcpyDeviceToHost));
    if (std::memcmp(c1v.data(), c2v.data(), c1v.size() * 4) != 0) return 1;
    std::printf("gather hint: columns of degree >= %d kept (%lld of %lld), 256-column product unchanged\n", min_degree,
                (long long)kept_cols, (long long)n_cols);
  }

  // ---- sub-matrix of flagged columns: count, fill, and the product with an operand that is
  //      zero in the other rows ----
  {
    std::vector<uint8_t> flag((size_t)n_cols);
    for (int64_t c2 = 0; c2 < n_cols; ++c2) flag[c2] = (uint8_t)(c2 % 7 == 0);
    uint8_t *d_flag = dev_alloc<uint8_t>((size_t)n_cols);
    int64_t *d_rp2 = dev_alloc<int64_t>((size_t)n_rows + 1);
    const size_t fb = gcnx_csr_filter_workspace_bytes(nnz);
    void *f_ws = dev_alloc<char>(fb);
    if (!d_flag || !d_rp2 || !f_ws) return 2;
    HIP_OK(hipMemcpy(d_flag, flag.data(), flag.size(), hipMemcpyHostToDevice));
    int64_t kept = -1;
    GCNX_OK_(gcnx_csr_filter_columns_count(&a, d_flag, d_rp2, &kept, f_ws, fb, stream));
    int64_t want_kept = 0;
    for (int64_t k = 0; k < nnz; ++k) want_kept += flag[col[k]];
    if (kept != want_kept || kept <= 0) return 1;
    int32_t *d_ci2 = dev_alloc<int32_t>((size_t)kept);
    float *d_v2 = dev_alloc<float>((size_t)kept);
    if (!d_ci2 || !d_v2) return 2;
    // a count that is not the one the workspace holds is refused
    if (gcnx_csr_filter_columns_fill(&a, f_ws, fb, kept - 1, d_ci2, d_v2, nullptr, stream) != GCNX_ERR_INVALID_ARG) return 1;
    GCNX_OK_(gcnx_csr_filter_columns_fill(&a, f_ws, fb, kept, d_ci2, d_v2, nullptr, stream));
    std::vector<int32_t> ci2((size_t)kept);
    HIP_OK(hipStreamSynchronize(stream));
    HIP_OK(hipMemcpy(ci2.data(), d_ci2, ci2.size() * 4, hipMemcpyDeviceToHost));
    for (int64_t k = 0; k < kept; ++k)
      if (!flag[ci2[k]]) return 1;
    gcnx_csr sub = a;
    sub.nnz = kept;
    sub.row_ptr = d_rp2;
    sub.col_idx = d_ci2;
    sub.val = d_v2;
    // operand: the rows of b outside the flagged set zeroed
    std::vector<float> bz(b);
    for (int64_t r2 = 0; r2 < n_cols; ++r2)
      if (!flag[r2])
        for (int j = 0; j < width; ++j) bz[(size_t)r2 * width + j] = 0.0f;
    float *d_bz = dev_alloc<float>(bz.size());
    float *d_cf = dev_alloc<float>((size_t)n_rows * width), *d_cs = dev_alloc<float>((size_t)n_rows * width);
    int32_t *d_chunk2 = dev_alloc<int32_t>((size_t)gcnx_spmm_num_chunks(kept) + 1);
    if (!d_bz || !d_cf || !d_cs || !d_chunk2) return 2;
    HIP_OK(hipMemcpy(d_bz, bz.data(), bz.size() * 4, hipMemcpyHostToDevice));
    GCNX_OK_(gcnx_spmm_chunk_table(&sub, d_chunk2, stream));
    GCNX_OK_(gcnx_spmm_csr_f32(&a, d_chunk, d_bz, width, d_cf, width, width, 1.0f, 0.0f, nullptr, nullptr, 1.0f, 0,
                               sp_ws, sp_bytes, stream));
    GCNX_OK_(gcnx_spmm_csr_f32(&sub, d_chunk2, d_bz, width, d_cs, width, width, 1.0f, 0.0f, nullptr, nullptr, 1.0f, 0,
                               sp_ws, sp_bytes, stream));
    std::vector<float> cf((size_t)n_rows * width), cs((size_t)n_rows * width);
    HIP_OK(hipStreamSynchronize(stream));
    HIP_OK(hipMemcpy(cf.data(), d_cf, cf.size() * 4, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(cs.data(), d_cs, cs.size() * 4, hipMemcpyDeviceToHost));
    double top = 0.0, apart = 0.0;
    for (size_t i = 0; i < cf.size(); ++i) {
      top = std::fabs((double)cf[i]) > top ? std::fabs((double)cf[i]) : top;
      const double d = std::fabs((double)cf[i] - (double)cs[i]);
      apart = d > apart ? d : apart;
    }
    std::printf("column filter: %lld of %lld nonzeros kept, products %.3g apart (largest entry %.3g)\n",
                (long long)kept, (long long)nnz, apart, top);
    if (!(apart <= 1e-6 * top)) return 1;
  }

  // ---- output dropout drawn by row id (gcnx_spmm_csr_rows_f32) ------------------
  {
    std::vector<int32_t> ids((size_t)n_rows), rev((size_t)n_rows);
    for (int64_t r2 = 0; r2 < n_rows; ++r2) {
      ids[(size_t)r2] = (int32_t)r2;
      rev[(size_t)r2] = (int32_t)(n_rows - 1 - r2);
    }
    int32_t *d_ids = dev_alloc<int32_t>(ids.size()), *d_rev = dev_alloc<int32_t>(rev.size());
    float *d_o1 = dev_alloc<float>((size_t)n_rows * width), *d_o2 = dev_alloc<float>((size_t)n_rows * width);
    float *d_o3 = dev_alloc<float>((size_t)n_rows * width);
    if (!d_ids || !d_rev || !d_o1 || !d_o2 || !d_o3) return 2;
    HIP_OK(hipMemcpy(d_ids, ids.data(), ids.size() * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_rev, rev.data(), rev.size() * 4, hipMemcpyHostToDevice));
    const gcnx_dropout drop = {0.5f, 99ull, 4ull * 1000ull, (int64_t)width, 0};
    GCNX_OK_(gcnx_spmm_csr_split_f32(&a, d_chunk, d_b, width, nullptr, 0, n_cols, d_o1, width, width, 1.0f, 0.0f, nullptr,
                                     nullptr, 1.0f, 1, &drop, sp_ws, sp_bytes, stream));
    GCNX_OK_(gcnx_spmm_csr_rows_f32(&a, d_chunk, d_b, width, nullptr, 0, n_cols, d_o2, width, width, 1.0f, 0.0f, 1, &drop,
                                    d_ids, nullptr, 0, sp_ws, sp_bytes, stream));
    GCNX_OK_(gcnx_spmm_csr_rows_f32(&a, d_chunk, d_b, width, nullptr, 0, n_cols, d_o3, width, width, 1.0f, 0.0f, 1, &drop,
                                    d_rev, nullptr, 0, sp_ws, sp_bytes, stream));
    std::vector<float> o1((size_t)n_rows * width), o2(o1.size()), o3(o1.size());
    HIP_OK(hipStreamSynchronize(stream));
    HIP_OK(hipMemcpy(o1.data(), d_o1, o1.size() * 4, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(o2.data(), d_o2, o2.size() * 4, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(o3.data(), d_o3, o3.size() * 4, hipMemcpyDeviceToHost));
    size_t differ = 0;
    for (size_t i = 0; i < o1.size(); ++i) {
      if (std::memcmp(&o1[i], &o2[i], 4) != 0) return 1;      // row r drawing the mask of row r: the plain epilogue
      differ += (o1[i] != 0.0f) != (o3[i] != 0.0f);
    }
    std::printf("dropout by row id: identity ids reproduce the plain epilogue; reversed ids change %zu of %zu elements\n",
                differ, o1.size());
    if (differ == 0) return 1;
  }

  // ---- dense product ---------------------------------------------------------
  std::vector<float> ha((size_t)M * K), hb((size_t)K * N), hc((size_t)M * N);
  for (auto &v : ha) v = (float)urand();
  for (auto &v : hb) v = (float)urand();
  float *d_ga = dev_alloc<float>(ha.size()), *d_gb = dev_alloc<float>(hb.size()), *d_gc = dev_alloc<float>(hc.size());
  HIP_OK(hipMemcpy(d_ga, ha.data(), ha.size() * 4, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(d_gb, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
  size_t g_bytes = gcnx_gemm_workspace_bytes(0, 0, M, N, K);
  void *g_ws = dev_alloc<char>(g_bytes);
  GCNX_OK_(gcnx_gemm_f32(0, 0, M, N, K, d_ga, K, d_gb, N, d_gc, N, 1.0f, 0.0f, 0, g_ws, g_bytes, stream));
  HIP_OK(hipStreamSynchronize(stream));
  HIP_OK(hipMemcpy(hc.data(), d_gc, hc.size() * 4, hipMemcpyDeviceToHost));
  worst = 0.0;
  for (int64_t m = 0; m < M; ++m)
    for (int64_t n = 0; n < N; ++n) {
      double s = 0.0, mg = 0.0;
      for (int64_t k = 0; k < K; ++k) {
        const double t = (double)ha[m * K + k] * (double)hb[k * N + n];
        s += t;
        mg += std::fabs(t);
      }
      const double err = std::fabs((double)hc[m * N + n] - s) / mg;
      if (err > worst) worst = err;
    }
  std::printf("gemm: worst error %.3g of the summed magnitudes\n", worst);
  if (!(worst <= 1e-5)) return 1;
  // ---- backward product with the factored operand ------------------------------
  // C = A * G,  G[c, j] = bit(c, j) ? scale * sum_k D[c, k] * W2[j, k] : 0, G never stored
  {
    const int32_t fw = 256, inner = 16;
    const int64_t words = fw / 32;
    if (!gcnx_spmm_factored_supported(fw, inner)) return 1;
    std::vector<float> hd((size_t)n_cols * inner), hw((size_t)fw * inner);
    std::vector<uint32_t> hbits((size_t)n_cols * words);
    for (auto &v : hd) v = (float)urand();
    for (auto &v : hw) v = (float)urand() * 0.3f;
    for (auto &v : hbits) v = rnd() & rnd();            // a quarter of the bits set
    float *d_d = dev_alloc<float>(hd.size()), *d_w = dev_alloc<float>(hw.size());
    float *d_fc = dev_alloc<float>((size_t)n_rows * fw);
    uint32_t *d_bits = dev_alloc<uint32_t>(hbits.size());
    if (!d_d || !d_w || !d_fc || !d_bits) return 2;
    HIP_OK(hipMemcpy(d_d, hd.data(), hd.size() * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_bits, hbits.data(), hbits.size() * 4, hipMemcpyHostToDevice));
    const size_t f_bytes = gcnx_spmm_workspace_bytes(&a, fw);
    void *f_ws = dev_alloc<char>(f_bytes);
    GCNX_OK_(gcnx_spmm_csr_factored_f32(&a, d_chunk, d_d, inner, d_bits, words, nullptr, 0, nullptr, 0, n_cols,
                                        d_w, inner, 1, inner, 2.0f, d_fc, fw, fw, f_ws, f_bytes, stream));
    std::vector<float> fc((size_t)n_rows * fw);
    HIP_OK(hipStreamSynchronize(stream));
    HIP_OK(hipMemcpy(fc.data(), d_fc, fc.size() * 4, hipMemcpyDeviceToHost));
    std::vector<double> g((size_t)n_cols * fw), gm((size_t)n_cols * fw);
    for (int64_t c0 = 0; c0 < n_cols; ++c0)
      for (int j = 0; j < fw; ++j) {
        double sum = 0.0, mg = 0.0;
        if ((hbits[(size_t)c0 * words + j / 32] >> (j % 32)) & 1u)
          for (int k = 0; k < inner; ++k) {
            const double t = 2.0 * (double)hd[(size_t)c0 * inner + k] * (double)hw[(size_t)j * inner + k];
            sum += t;
            mg += std::fabs(t);
          }
        g[(size_t)c0 * fw + j] = sum;
        gm[(size_t)c0 * fw + j] = mg;
      }
    std::vector<double> fref((size_t)n_rows * fw, 0.0), fmag((size_t)n_rows * fw, 0.0);
    for (int64_t k = 0; k < nnz; ++k) {
      const double v = (double)(float)val[k];
      for (int j = 0; j < fw; ++j) {
        fref[(size_t)row[k] * fw + j] += v * g[(size_t)col[k] * fw + j];
        fmag[(size_t)row[k] * fw + j] += std::fabs(v) * gm[(size_t)col[k] * fw + j];
      }
    }
    worst = 0.0;
    for (size_t i = 0; i < fref.size(); ++i) {
      const double err = std::fabs((double)fc[i] - fref[i]) / (fmag[i] > 1e-30 ? fmag[i] : 1e-30);
      if (fmag[i] > 1e-30 && err > worst) worst = err;
      if (fmag[i] <= 1e-30 && fc[i] != 0.0f) return 1;
    }
    std::printf("factored spmm: worst error %.3g of the summed magnitudes\n", worst);
    if (!(worst <= 1e-5)) return 1;
    // an inner size the kernel does not serve is a status, not a fault
    if (gcnx_spmm_csr_factored_f32(&a, d_chunk, d_d, 8, d_bits, words, nullptr, 0, nullptr, 0, n_cols, d_w, 8, 1, 8,
                                   2.0f, d_fc, fw, fw, f_ws, f_bytes, stream) != GCNX_ERR_UNSUPPORTED) return 1;
  }
  // ---- halo exchange over RCCL, driven from this process alone ------------------
  // One rank (RCCL refuses two ranks on one GPU): the id, the communicator, the
  // packed send and the grouped send / receive are the calls a multi-process host
  // makes, here with this rank as its own peer.
  if (gcnx_comm_available()) {
    char id[GCNX_COMM_ID_BYTES];
    void *comm = nullptr;
    GCNX_OK_(gcnx_comm_unique_id(id));
    GCNX_OK_(gcnx_comm_create(1, 0, id, &comm));
    const int64_t n_ask = 1500;
    std::vector<int32_t> ask(n_ask);
    for (auto &v : ask) v = (int32_t)(rnd() % n_cols);
    int32_t *d_ask = dev_alloc<int32_t>(n_ask);
    float *d_send = dev_alloc<float>((size_t)n_ask * 16), *d_halo = dev_alloc<float>((size_t)n_ask * 16);
    if (!d_ask || !d_send || !d_halo) return 2;
    HIP_OK(hipMemcpy(d_ask, ask.data(), n_ask * 4, hipMemcpyHostToDevice));
    const int64_t rows[1] = {n_ask};
    gcnx_halo_exchange plan = {1, 0, rows, rows, d_ask};
    // columns [32, 48) of the rows of b that "the peer" asked for
    GCNX_OK_(gcnx_halo_allgather_f32(&plan, comm, d_b, width, 32, 16, d_send, d_halo, stream));
    float one_val = 3.0f, *d_one = dev_alloc<float>(1);
    HIP_OK(hipMemcpyAsync(d_one, &one_val, 4, hipMemcpyHostToDevice, stream));
    GCNX_OK_(gcnx_comm_allreduce_sum_f32(comm, d_one, 1, stream));
    std::vector<float> halo((size_t)n_ask * 16);
    HIP_OK(hipStreamSynchronize(stream));
    HIP_OK(hipMemcpy(halo.data(), d_halo, halo.size() * 4, hipMemcpyDeviceToHost));
    HIP_OK(hipMemcpy(&one_val, d_one, 4, hipMemcpyDeviceToHost));
    for (int64_t j = 0; j < n_ask; ++j)
      for (int cidx = 0; cidx < 16; ++cidx)
        if (halo[(size_t)j * 16 + cidx] != b[(size_t)ask[j] * width + 32 + cidx]) return 1;
    if (one_val != 3.0f) return 1;
    GCNX_OK_(gcnx_comm_destroy(comm));
    std::printf("halo exchange over RCCL (1 rank): %lld rows x 16 columns delivered\n", (long long)n_ask);
  } else {
    std::printf("RCCL not available here: halo exchange skipped (%s)\n", gcnx_last_error());
  }
  std::printf("C ABI client OK (abi %d)\n", gcnx_abi_version());
  return 0;
}
