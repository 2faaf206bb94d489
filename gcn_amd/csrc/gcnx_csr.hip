// Device-side CSR construction: COO -> CSR, exact transpose, and key-list -> CSR
// for the synthetic graphs.  Ordering is done with one stable LSD radix sort of
// 64-bit (major << 32 | minor) keys (rocPRIM); everything around it is plain
// HIP.  No arithmetic touches the values except the final fp64 -> fp32 cast.
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "gcnx_common.h"

namespace gcnx {

__global__ void coo_keys_kernel(int64_t nnz, int64_t n_rows, int64_t n_cols,
                                const int32_t *__restrict__ row, const int32_t *__restrict__ col,
                                int64_t stride, uint64_t *__restrict__ keys,
                                int64_t *__restrict__ idx, int *__restrict__ bad) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= nnz) return;
  const int32_t r = row[k * stride];
  const int32_t c = col[k * stride];
  if (r < 0 || r >= n_rows || c < 0 || c >= n_cols) {
    atomicOr(bad, 1);
    keys[k] = 0;
  } else {
    keys[k] = ((uint64_t)(uint32_t)r << 32) | (uint32_t)c;
  }
  idx[k] = k;
}

// row of slot s = last r with row_ptr[r] <= s
__device__ __forceinline__ int64_t row_of_slot(const int64_t *row_ptr, int64_t n_rows, int64_t s) {
  int64_t lo = 0, hi = n_rows;
  while (hi - lo > 1) {
    const int64_t mid = lo + (hi - lo) / 2;
    if (row_ptr[mid] <= s) lo = mid; else hi = mid;
  }
  return lo;
}

__global__ void transpose_keys_kernel(int64_t nnz, int64_t n_rows,
                                      const int64_t *__restrict__ row_ptr,
                                      const int32_t *__restrict__ col_idx,
                                      uint64_t *__restrict__ keys, int64_t *__restrict__ idx) {
  const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= nnz) return;
  const int64_t r = row_of_slot(row_ptr, n_rows, s);
  keys[s] = ((uint64_t)(uint32_t)col_idx[s] << 32) | (uint32_t)r;
  idx[s] = s;
}

// After sorting: minor index, permuted value (cast last) and permutation.
__global__ void csr_fill_kernel(int64_t nnz, const uint64_t *__restrict__ keys,
                                const int64_t *__restrict__ idx, const void *__restrict__ val,
                                int val_is_f64, int32_t *__restrict__ col_idx,
                                float *__restrict__ val_out, int64_t *__restrict__ perm) {
  const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= nnz) return;
  const int64_t k = idx[s];
  col_idx[s] = (int32_t)(uint32_t)(keys[s] & 0xffffffffull);
  if (val) {
    val_out[s] = val_is_f64 ? (float)static_cast<const double *>(val)[k]
                            : static_cast<const float *>(val)[k];
  }
  if (perm) perm[s] = k;
}

// row_ptr[r] = number of sorted keys whose major index is < r (+ base row).
__global__ void row_ptr_kernel(int64_t n_rows, int64_t row_base, int64_t nnz,
                               const uint64_t *__restrict__ keys, int64_t *__restrict__ row_ptr) {
  const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (r > n_rows) return;
  const uint64_t target = (uint64_t)(r + row_base) << 32;
  int64_t lo = 0, hi = nnz;  // first position with key >= target
  while (lo < hi) {
    const int64_t mid = lo + (hi - lo) / 2;
    if (keys[mid] < target) lo = mid + 1; else hi = mid;
  }
  row_ptr[r] = lo;
}

static int bits_for(int64_t n) {
  int b = 1;
  while (b < 63 && (1ll << b) < n) ++b;
  return b;
}

static size_t sort_pairs_temp_bytes(int64_t n, int end_bit) {
  size_t bytes = 0;
  uint64_t *k = nullptr;
  int64_t *v = nullptr;
  (void)rocprim::radix_sort_pairs(nullptr, bytes, k, k, v, v, (size_t)n, 0u, (unsigned)end_bit,
                                  (hipStream_t)0);
  return bytes;
}

static size_t sort_keys_temp_bytes(int64_t n, int end_bit) {
  size_t bytes = 0;
  uint64_t *k = nullptr;
  (void)rocprim::radix_sort_keys(nullptr, bytes, k, k, (size_t)n, 0u, (unsigned)end_bit,
                                 (hipStream_t)0);
  return bytes;
}

struct PairLayout {
  size_t keys_in, keys_out, idx_in, idx_out, flag, temp, temp_bytes, total;
};

static PairLayout pair_layout(int64_t nnz, int end_bit) {
  PairLayout l;
  const size_t n = (size_t)(nnz > 0 ? nnz : 1);
  size_t off = 0;
  l.keys_in = off;  off += align_up(n * 8, 256);
  l.keys_out = off; off += align_up(n * 8, 256);
  l.idx_in = off;   off += align_up(n * 8, 256);
  l.idx_out = off;  off += align_up(n * 8, 256);
  l.flag = off;     off += 256;
  l.temp = off;
  l.temp_bytes = nnz > 0 ? sort_pairs_temp_bytes(nnz, end_bit) : 0;
  off += align_up(l.temp_bytes, 256);
  l.total = off;
  return l;
}

// keys/idx -> sorted -> (row_ptr, col_idx, val, perm)
static int finish_from_pairs(const PairLayout &l, char *ws, int64_t n_major, int64_t nnz,
                             int end_bit, const void *val, int val_is_f64, int64_t *row_ptr,
                             int32_t *col_idx, float *val_out, int64_t *perm, hipStream_t st) {
  uint64_t *keys_in = reinterpret_cast<uint64_t *>(ws + l.keys_in);
  uint64_t *keys_out = reinterpret_cast<uint64_t *>(ws + l.keys_out);
  int64_t *idx_in = reinterpret_cast<int64_t *>(ws + l.idx_in);
  int64_t *idx_out = reinterpret_cast<int64_t *>(ws + l.idx_out);
  if (nnz > 0) {
    size_t tb = l.temp_bytes;
    GCNX_CHECK_HIP(rocprim::radix_sort_pairs(ws + l.temp, tb, keys_in, keys_out, idx_in, idx_out,
                                             (size_t)nnz, 0u, (unsigned)end_bit, st));
    hipLaunchKernelGGL(csr_fill_kernel, dim3((unsigned)ceil_div(nnz, 256)), dim3(256), 0, st, nnz,
                       keys_out, idx_out, val, val_is_f64, col_idx, val_out, perm);
    GCNX_CHECK_LAUNCH();
  }
  hipLaunchKernelGGL(row_ptr_kernel, dim3((unsigned)ceil_div(n_major + 1, 256)), dim3(256), 0, st,
                     n_major, (int64_t)0, nnz, keys_out, row_ptr);
  GCNX_CHECK_LAUNCH();
  return GCNX_OK;
}

// ---- key list -> CSR (synthetic graphs) -----------------------------------

__global__ void append_diag_kernel(int64_t row_begin, int64_t n_local, uint64_t *__restrict__ keys) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_local) return;
  const uint64_t r = (uint64_t)(row_begin + i);
  keys[i] = (r << 32) | r;
}

__global__ void unique_flag_kernel(int64_t n, const uint64_t *__restrict__ keys,
                                   int64_t *__restrict__ flag) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  flag[i] = (i == 0 || keys[i] != keys[i - 1]) ? 1 : 0;
}

__global__ void unique_scatter_kernel(int64_t n, const uint64_t *__restrict__ keys,
                                      const int64_t *__restrict__ pos,
                                      uint64_t *__restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (i == 0 || keys[i] != keys[i - 1]) out[pos[i]] = keys[i];
}

__global__ void keys_to_csr_kernel(int64_t n, const uint64_t *__restrict__ keys,
                                   int32_t *__restrict__ col_idx, float *__restrict__ val) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  col_idx[i] = (int32_t)(uint32_t)(keys[i] & 0xffffffffull);
  val[i] = 1.0f;
}

struct KeyLayout {
  size_t keys_sorted, pos, count, temp, temp_bytes, total;
};

static KeyLayout key_layout(int64_t n_keys) {
  KeyLayout l;
  const size_t n = (size_t)(n_keys > 0 ? n_keys : 1);
  size_t off = 0;
  l.keys_sorted = off; off += align_up(n * 8, 256);
  l.pos = off;         off += align_up(n * 8, 256);
  l.count = off;       off += 256;
  l.temp = off;
  size_t t1 = n_keys > 0 ? sort_keys_temp_bytes(n_keys, 64) : 0;
  size_t t2 = 0;
  int64_t *p = nullptr;
  (void)rocprim::exclusive_scan(nullptr, t2, p, p, (int64_t)0, n, rocprim::plus<int64_t>(),
                                (hipStream_t)0);
  l.temp_bytes = t1 > t2 ? t1 : t2;
  off += align_up(l.temp_bytes, 256);
  l.total = off;
  return l;
}

// ---- halo plan -------------------------------------------------------------
__global__ void halo_mark_kernel(int64_t nnz, const int32_t *__restrict__ col_idx, int64_t lo,
                                 int64_t hi, int32_t *__restrict__ flag) {
  const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= nnz) return;
  const int64_t c = col_idx[s];
  if (c < lo || c >= hi) flag[c] = 1;  // same value from every writer
}

__global__ void halo_compact_kernel(int64_t n, const int32_t *__restrict__ flag,
                                    const int32_t *__restrict__ pos, int64_t cap,
                                    int32_t *__restrict__ need) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n) return;
  if (flag[c] && pos[c] < cap) need[pos[c]] = (int32_t)c;
}

__global__ void halo_remap_kernel(int64_t nnz, const int32_t *__restrict__ col_idx, int64_t lo,
                                  int64_t hi, const int32_t *__restrict__ pos,
                                  int32_t *__restrict__ out) {
  const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= nnz) return;
  const int64_t c = col_idx[s];
  out[s] = (c >= lo && c < hi) ? (int32_t)(c - lo) : (int32_t)((hi - lo) + pos[c]);
}

// one wavefront per 4 rows of 16-byte pieces when aligned, else scalar
__global__ __launch_bounds__(kBlock) void gather_rows_kernel(int64_t n_idx,
                                                            const int32_t *__restrict__ idx,
                                                            const float *__restrict__ src,
                                                            int64_t lds, int32_t width,
                                                            float *__restrict__ dst, int64_t ldd,
                                                            int vec) {
  typedef float __attribute__((ext_vector_type(4))) f4;
  const int per_row = vec ? width / 4 : width;
  const int64_t total = n_idx * (int64_t)per_row;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t j = e / per_row;
    const int q = (int)(e % per_row);
    const int64_t r = idx[j];
    if (vec)
      reinterpret_cast<f4 *>(dst + j * ldd)[q] = reinterpret_cast<const f4 *>(src + r * lds)[q];
    else
      dst[j * ldd + q] = src[r * lds + q];
  }
}

// the inverse placement: dst[idx[j], :] = src[j, :] (distinct idx: no two rows collide)
__global__ __launch_bounds__(kBlock) void scatter_rows_kernel(int64_t n_idx,
                                                             const int32_t *__restrict__ idx,
                                                             const float *__restrict__ src,
                                                             int64_t lds, int32_t width,
                                                             float *__restrict__ dst, int64_t ldd,
                                                             int vec) {
  typedef float __attribute__((ext_vector_type(4))) f4;
  const int per_row = vec ? width / 4 : width;
  const int64_t total = n_idx * (int64_t)per_row;
  for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total;
       e += (int64_t)gridDim.x * blockDim.x) {
    const int64_t j = e / per_row;
    const int q = (int)(e % per_row);
    const int64_t r = idx[j];
    if (vec)
      reinterpret_cast<f4 *>(dst + r * ldd)[q] = reinterpret_cast<const f4 *>(src + j * lds)[q];
    else
      dst[r * ldd + q] = src[j * lds + q];
  }
}

__global__ __launch_bounds__(kBlock) void column_degree_kernel(int64_t nnz, const int32_t *__restrict__ col_idx,
                                                              int32_t *__restrict__ degree) {
  for (int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; k < nnz; k += (int64_t)gridDim.x * blockDim.x)
    atomicAdd(degree + col_idx[k], 1);     // integer counts: the order of the additions does not matter
}

// how many columns have degree d, for d < kDegreeBins - 1; the last bin collects the rest
constexpr int kDegreeBins = 4096;
__global__ __launch_bounds__(kBlock) void degree_histogram_kernel(int64_t n, const int32_t *__restrict__ degree,
                                                                 int32_t *__restrict__ bins) {
  __shared__ int32_t local[kDegreeBins];
  for (int i = threadIdx.x; i < kDegreeBins; i += blockDim.x) local[i] = 0;
  __syncthreads();
  for (int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; c < n; c += (int64_t)gridDim.x * blockDim.x) {
    const int32_t d = degree[c];
    atomicAdd(local + (d < kDegreeBins - 1 ? (d < 0 ? 0 : d) : kDegreeBins - 1), 1);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < kDegreeBins; i += blockDim.x)
    if (local[i]) atomicAdd(bins + i, local[i]);
}

// one wavefront per 64 nonzeros: the ballot of (degree of my column >= min_degree) IS the word
__global__ __launch_bounds__(kBlock) void gather_keep_kernel(int64_t nnz, const int32_t *__restrict__ col_idx,
                                                            const int32_t *__restrict__ degree, int32_t min_degree,
                                                            unsigned long long *__restrict__ keep) {
  const int64_t words = (nnz + 63) / 64;
  const int lane = threadIdx.x & 63;
  for (int64_t w = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6; w < words;
       w += ((int64_t)gridDim.x * blockDim.x) >> 6) {
    const int64_t k = w * 64 + lane;
    const bool hot = k < nnz && degree[col_idx[k]] >= min_degree;
    const unsigned long long bits = __ballot(hot);
    if (lane == 0) keep[w] = bits;
  }
}

// ---- sub-matrix by column flags ---------------------------------------------
// one wavefront per 64 nonzeros: the ballot of (my column is kept) is the keep word,
// its population count the word's share of the output
__global__ __launch_bounds__(kBlock) void filter_mark_kernel(int64_t nnz, int64_t n_cols,
                                                            const int32_t *__restrict__ col_idx,
                                                            const uint8_t *__restrict__ col_keep,
                                                            unsigned long long *__restrict__ words,
                                                            int64_t *__restrict__ count) {
  const int64_t n_words = (nnz + 63) / 64;
  const int lane = threadIdx.x & 63;
  for (int64_t w = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6; w < n_words;
       w += ((int64_t)gridDim.x * blockDim.x) >> 6) {
    const int64_t k = w * 64 + lane;
    bool on = false;
    if (k < nnz) {
      const int64_t c = col_idx[k];
      on = c >= 0 && c < n_cols && col_keep[c] != 0;
    }
    const unsigned long long bits = __ballot(on);
    if (lane == 0) {
      words[w] = bits;
      count[w] = __popcll(bits);
    }
  }
}

// number of kept nonzeros in slots [0, k); before[] is the exclusive scan of the words' counts
__device__ __forceinline__ int64_t kept_before(int64_t k, const unsigned long long *__restrict__ words,
                                               const int64_t *__restrict__ before) {
  const int64_t w = k >> 6;
  const int b = (int)(k & 63);
  int64_t r = before[w];
  if (b) r += __popcll(words[w] & ((1ull << b) - 1ull));
  return r;
}

__global__ void filter_row_ptr_kernel(int64_t n_rows, int64_t nnz, const int64_t *__restrict__ row_ptr,
                                      const unsigned long long *__restrict__ words,
                                      const int64_t *__restrict__ before, int64_t *__restrict__ out_row_ptr) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i > n_rows) return;
  int64_t k = row_ptr[i];
  k = k < 0 ? 0 : (k > nnz ? nnz : k);
  out_row_ptr[i] = kept_before(k, words, before);
}

// kept nonzeros move up in order: slot k goes to before[word] + (kept slots below it in its word)
__global__ __launch_bounds__(kBlock) void filter_fill_kernel(int64_t nnz, const int32_t *__restrict__ col_idx,
                                                            const float *__restrict__ val,
                                                            const unsigned long long *__restrict__ words,
                                                            const int64_t *__restrict__ before,
                                                            int32_t *__restrict__ out_col, float *__restrict__ out_val,
                                                            int64_t *__restrict__ out_src) {
  const int64_t n_words = (nnz + 63) / 64;
  const int lane = threadIdx.x & 63;
  for (int64_t w = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6; w < n_words;
       w += ((int64_t)gridDim.x * blockDim.x) >> 6) {
    const unsigned long long bits = words[w];
    if ((bits >> lane) & 1ull) {
      const int64_t k = w * 64 + lane;
      const int64_t pos = before[w] + __popcll(bits & ((1ull << lane) - 1ull));
      out_col[pos] = col_idx[k];
      out_val[pos] = val[k];
      if (out_src) out_src[pos] = k;
    }
  }
}

}  // namespace gcnx

using namespace gcnx;

static size_t halo_scan_temp(int64_t n) {
  size_t t = 0;
  int32_t *p = nullptr;
  (void)rocprim::exclusive_scan(nullptr, t, p, p, (int32_t)0, (size_t)(n > 0 ? n : 1),
                                rocprim::plus<int32_t>(), (hipStream_t)0);
  return t;
}

extern "C" size_t gcnx_halo_plan_workspace_bytes(int64_t n_cols_global) {
  if (n_cols_global < 0) return 0;
  const size_t n = (size_t)(n_cols_global > 0 ? n_cols_global : 1);
  return 2 * align_up(n * 4, 256) + align_up(halo_scan_temp(n_cols_global), 256) + 256;
}

extern "C" int gcnx_halo_plan(const gcnx_csr *a, int64_t row_begin, int64_t n_local,
                              int32_t *col_remapped, int32_t *need, int64_t need_cap,
                              int64_t *h_n_need, void *ws, size_t ws_bytes, void *stream) {
  GCNX_REQUIRE(a && h_n_need && row_begin >= 0 && n_local >= 0 && need_cap >= 0,
               GCNX_ERR_INVALID_ARG, "halo_plan: bad argument");
  GCNX_REQUIRE(a->n_cols < INT32_MAX && row_begin + n_local <= a->n_cols, GCNX_ERR_INVALID_ARG,
               "halo_plan: bad column range");
  GCNX_REQUIRE(a->nnz == 0 || (a->col_idx && col_remapped), GCNX_ERR_INVALID_ARG,
               "halo_plan: null array");
  const size_t needws = gcnx_halo_plan_workspace_bytes(a->n_cols);
  GCNX_REQUIRE(ws && ws_bytes >= needws, GCNX_ERR_WORKSPACE,
               "halo_plan: workspace of %zu bytes required, got %zu", needws, ws_bytes);
  hipStream_t st = as_stream(stream);
  const int64_t n = a->n_cols;
  char *w = static_cast<char *>(ws);
  const size_t arr = align_up((size_t)(n > 0 ? n : 1) * 4, 256);
  int32_t *flag = reinterpret_cast<int32_t *>(w);
  int32_t *pos = reinterpret_cast<int32_t *>(w + arr);
  void *temp = w + 2 * arr;
  *h_n_need = 0;
  if (n == 0) return GCNX_OK;
  GCNX_CHECK_HIP(hipMemsetAsync(flag, 0, (size_t)n * 4, st));
  const int64_t lo = row_begin, hi = row_begin + n_local;
  if (a->nnz > 0) {
    hipLaunchKernelGGL(halo_mark_kernel, dim3((unsigned)ceil_div(a->nnz, 256)), dim3(256), 0, st,
                       a->nnz, a->col_idx, lo, hi, flag);
    GCNX_CHECK_LAUNCH();
  }
  size_t tb = halo_scan_temp(n);
  GCNX_CHECK_HIP(rocprim::exclusive_scan(temp, tb, flag, pos, (int32_t)0, (size_t)n,
                                         rocprim::plus<int32_t>(), st));
  int32_t last_pos = 0, last_flag = 0;
  GCNX_CHECK_HIP(hipMemcpyAsync(&last_pos, pos + n - 1, 4, hipMemcpyDeviceToHost, st));
  GCNX_CHECK_HIP(hipMemcpyAsync(&last_flag, flag + n - 1, 4, hipMemcpyDeviceToHost, st));
  GCNX_CHECK_HIP(hipStreamSynchronize(st));
  const int64_t n_need = (int64_t)last_pos + last_flag;
  *h_n_need = n_need;
  GCNX_REQUIRE(n_need <= need_cap && (n_need == 0 || need), GCNX_ERR_WORKSPACE,
               "halo_plan: %lld remote rows referenced but need[] holds %lld", (long long)n_need,
               (long long)need_cap);
  GCNX_REQUIRE(n_local + n_need < INT32_MAX, GCNX_ERR_UNSUPPORTED, "halo_plan: too many columns");
  if (n_need > 0) {
    hipLaunchKernelGGL(halo_compact_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, st, n,
                       flag, pos, need_cap, need);
    GCNX_CHECK_LAUNCH();
  }
  if (a->nnz > 0) {
    hipLaunchKernelGGL(halo_remap_kernel, dim3((unsigned)ceil_div(a->nnz, 256)), dim3(256), 0, st,
                       a->nnz, a->col_idx, lo, hi, pos, col_remapped);
    GCNX_CHECK_LAUNCH();
  }
  GCNX_CHECK_HIP(hipStreamSynchronize(st));
  return GCNX_OK;
}

extern "C" int gcnx_gather_rows_f32(int64_t n_idx, const int32_t *idx, const float *src,
                                    int64_t lds, int32_t width, float *dst, int64_t ldd,
                                    void *stream) {
  GCNX_REQUIRE(n_idx >= 0 && width >= 0, GCNX_ERR_INVALID_ARG, "gather_rows: negative size");
  if (n_idx == 0 || width == 0) return GCNX_OK;
  GCNX_REQUIRE(idx && src && dst && lds >= width && ldd >= width, GCNX_ERR_INVALID_ARG,
               "gather_rows: bad argument");
  const int vec = (width % 4 == 0) && (lds % 4 == 0) && (ldd % 4 == 0) &&
                  ((reinterpret_cast<uintptr_t>(src) & 15) == 0) &&
                  ((reinterpret_cast<uintptr_t>(dst) & 15) == 0);
  const int64_t total = n_idx * (int64_t)(vec ? width / 4 : width);
  int64_t blocks = ceil_div(total, kBlock);
  if (blocks > 256 * 32) blocks = 256 * 32;
  hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, as_stream(stream),
                     n_idx, idx, src, lds, width, dst, ldd, vec);
  GCNX_CHECK_LAUNCH();
  return GCNX_OK;
}

extern "C" int gcnx_csr_column_degrees(const gcnx_csr *a, int32_t *degree, void *stream) {
  GCNX_REQUIRE(a && a->n_cols >= 0 && a->nnz >= 0, GCNX_ERR_INVALID_ARG, "column_degrees: bad matrix");
  if (a->n_cols == 0) return GCNX_OK;
  GCNX_REQUIRE(degree && (a->nnz == 0 || a->col_idx), GCNX_ERR_INVALID_ARG, "column_degrees: null array");
  hipStream_t st = as_stream(stream);
  GCNX_CHECK_HIP(hipMemsetAsync(degree, 0, (size_t)a->n_cols * sizeof(int32_t), st));
  if (a->nnz == 0) return GCNX_OK;
  int64_t blocks = ceil_div(a->nnz, kBlock);
  if (blocks > 256 * 32) blocks = 256 * 32;
  hipLaunchKernelGGL(column_degree_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, st, a->nnz, a->col_idx, degree);
  GCNX_CHECK_LAUNCH();
  return GCNX_OK;
}

extern "C" size_t gcnx_csr_degree_threshold_workspace_bytes(void) { return kDegreeBins * sizeof(int32_t); }

extern "C" int gcnx_csr_degree_threshold(const int32_t *degree, int64_t n_cols, int64_t max_columns,
                                         int32_t *h_min_degree, void *ws, size_t ws_bytes, void *stream) {
  GCNX_REQUIRE(n_cols >= 0 && max_columns >= 0 && h_min_degree, GCNX_ERR_INVALID_ARG, "degree_threshold: bad argument");
  *h_min_degree = 2;
  if (n_cols == 0) return GCNX_OK;
  GCNX_REQUIRE(degree, GCNX_ERR_INVALID_ARG, "degree_threshold: null degrees");
  GCNX_REQUIRE(ws && ws_bytes >= kDegreeBins * sizeof(int32_t), GCNX_ERR_WORKSPACE,
               "degree_threshold: workspace of %zu bytes required, got %zu", kDegreeBins * sizeof(int32_t), ws_bytes);
  hipStream_t st = as_stream(stream);
  int32_t *bins = static_cast<int32_t *>(ws);
  GCNX_CHECK_HIP(hipMemsetAsync(bins, 0, kDegreeBins * sizeof(int32_t), st));
  int64_t blocks = ceil_div(n_cols, (int64_t)kBlock * 16);
  if (blocks > 1024) blocks = 1024;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(degree_histogram_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, st, n_cols, degree, bins);
  GCNX_CHECK_LAUNCH();
  static thread_local int32_t host[kDegreeBins];
  GCNX_CHECK_HIP(hipMemcpyAsync(host, bins, sizeof(host), hipMemcpyDeviceToHost, st));
  GCNX_CHECK_HIP(hipStreamSynchronize(st));
  // the smallest d >= 2 with (columns of degree >= d) <= max_columns; degrees from
  // kDegreeBins - 1 on share the last bin, so the answer is capped there
  int64_t at_least = 0;
  int32_t d = kDegreeBins - 1;
  for (; d >= 2; --d) {
    if (at_least + host[d] > max_columns) break;
    at_least += host[d];
  }
  *h_min_degree = d + 1 > kDegreeBins - 1 ? kDegreeBins - 1 : d + 1;
  return GCNX_OK;
}

extern "C" int gcnx_csr_gather_keep(const gcnx_csr *a, const int32_t *degree, int32_t min_degree,
                                    uint64_t *keep, void *stream) {
  GCNX_REQUIRE(a && a->nnz >= 0, GCNX_ERR_INVALID_ARG, "gather_keep: bad matrix");
  if (a->nnz == 0) return GCNX_OK;
  GCNX_REQUIRE(a->col_idx && degree && keep, GCNX_ERR_INVALID_ARG, "gather_keep: null array");
  int64_t blocks = ceil_div(ceil_div(a->nnz, 64), kBlock / 64);
  if (blocks > 256 * 32) blocks = 256 * 32;
  hipLaunchKernelGGL(gather_keep_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, as_stream(stream), a->nnz,
                     a->col_idx, degree, min_degree, reinterpret_cast<unsigned long long *>(keep));
  GCNX_CHECK_LAUNCH();
  return GCNX_OK;
}

// ---- sub-matrix by column flags: workspace = keep words | scanned counts | scan scratch
namespace {
inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
struct FilterLayout {
  size_t words, before, temp, temp_bytes, total;
  int64_t n_scan;   // words + 1: the last entry of the scan is the number of kept nonzeros
};
FilterLayout filter_layout(int64_t nnz) {
  FilterLayout l;
  const int64_t n_words = (nnz + 63) / 64;
  l.n_scan = n_words + 1;
  const size_t arr = align_up((size_t)l.n_scan * 8, 256);
  l.words = 0;
  l.before = arr;
  l.temp = 2 * arr;
  size_t t = 0;
  int64_t *p = nullptr;
  (void)rocprim::exclusive_scan(nullptr, t, p, p, (int64_t)0, (size_t)l.n_scan, rocprim::plus<int64_t>(),
                                (hipStream_t)0);
  l.temp_bytes = t;
  l.total = l.temp + align_up(t, 256) + 256;
  return l;
}
}  // namespace

extern "C" size_t gcnx_csr_filter_workspace_bytes(int64_t nnz) { return nnz < 0 ? 0 : filter_layout(nnz).total; }

extern "C" int gcnx_csr_filter_columns_count(const gcnx_csr *a, const uint8_t *col_keep, int64_t *out_row_ptr,
                                             int64_t *h_nnz, void *ws, size_t ws_bytes, void *stream) {
  GCNX_REQUIRE(a && h_nnz && a->n_rows >= 0 && a->n_cols >= 0 && a->nnz >= 0, GCNX_ERR_INVALID_ARG,
               "filter_columns: bad argument");
  *h_nnz = 0;
  GCNX_REQUIRE(out_row_ptr, GCNX_ERR_INVALID_ARG, "filter_columns: null out_row_ptr");
  hipStream_t st = as_stream(stream);
  if (a->nnz == 0) {
    GCNX_CHECK_HIP(hipMemsetAsync(out_row_ptr, 0, (size_t)(a->n_rows + 1) * sizeof(int64_t), st));
    return GCNX_OK;
  }
  GCNX_REQUIRE(a->row_ptr && a->col_idx && col_keep, GCNX_ERR_INVALID_ARG, "filter_columns: null array");
  const FilterLayout l = filter_layout(a->nnz);
  GCNX_REQUIRE(ws && ws_bytes >= l.total && aligned16(ws), GCNX_ERR_WORKSPACE,
               "filter_columns: workspace of %zu bytes (16-byte aligned) required, got %zu", l.total, ws_bytes);
  char *w = static_cast<char *>(ws);
  unsigned long long *words = reinterpret_cast<unsigned long long *>(w + l.words);
  int64_t *before = reinterpret_cast<int64_t *>(w + l.before);
  const int64_t n_words = l.n_scan - 1;
  // the entry past the last word stays zero, so that the scan's last value is the total
  GCNX_CHECK_HIP(hipMemsetAsync(before + n_words, 0, sizeof(int64_t), st));
  GCNX_CHECK_HIP(hipMemsetAsync(words + n_words, 0, sizeof(unsigned long long), st));
  int64_t blocks = ceil_div(n_words, kBlock / 64);
  if (blocks > 256 * 32) blocks = 256 * 32;
  hipLaunchKernelGGL(filter_mark_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, st, a->nnz, a->n_cols, a->col_idx,
                     col_keep, words, before);
  GCNX_CHECK_LAUNCH();
  size_t tb = l.temp_bytes;
  GCNX_CHECK_HIP(rocprim::exclusive_scan(w + l.temp, tb, before, before, (int64_t)0, (size_t)l.n_scan,
                                         rocprim::plus<int64_t>(), st));
  hipLaunchKernelGGL(filter_row_ptr_kernel, dim3((unsigned)ceil_div(a->n_rows + 1, 256)), dim3(256), 0, st,
                     a->n_rows, a->nnz, a->row_ptr, words, before, out_row_ptr);
  GCNX_CHECK_LAUNCH();
  int64_t total = 0;
  GCNX_CHECK_HIP(hipMemcpyAsync(&total, before + n_words, sizeof(int64_t), hipMemcpyDeviceToHost, st));
  GCNX_CHECK_HIP(hipStreamSynchronize(st));
  *h_nnz = total;
  return GCNX_OK;
}

extern "C" int gcnx_csr_filter_columns_fill(const gcnx_csr *a, const void *ws, size_t ws_bytes, int64_t out_nnz,
                                            int32_t *out_col, float *out_val, int64_t *out_src, void *stream) {
  GCNX_REQUIRE(a && a->nnz >= 0 && out_nnz >= 0 && out_nnz <= a->nnz, GCNX_ERR_INVALID_ARG,
               "filter_columns_fill: bad argument");
  if (a->nnz == 0 || out_nnz == 0) return GCNX_OK;
  GCNX_REQUIRE(a->col_idx && a->val && out_col && out_val, GCNX_ERR_INVALID_ARG, "filter_columns_fill: null array");
  const FilterLayout l = filter_layout(a->nnz);
  GCNX_REQUIRE(ws && ws_bytes >= l.total && aligned16(ws), GCNX_ERR_WORKSPACE,
               "filter_columns_fill: the workspace gcnx_csr_filter_columns_count filled (%zu bytes) is required, got %zu",
               l.total, ws_bytes);
  hipStream_t st = as_stream(stream);
  const char *w = static_cast<const char *>(ws);
  const unsigned long long *words = reinterpret_cast<const unsigned long long *>(w + l.words);
  const int64_t *before = reinterpret_cast<const int64_t *>(w + l.before);
  const int64_t n_words = l.n_scan - 1;
  // the caller's arrays hold out_nnz entries: it must be the count the first call returned
  int64_t total = 0;
  GCNX_CHECK_HIP(hipMemcpyAsync(&total, before + n_words, sizeof(int64_t), hipMemcpyDeviceToHost, st));
  GCNX_CHECK_HIP(hipStreamSynchronize(st));
  GCNX_REQUIRE(total == out_nnz, GCNX_ERR_INVALID_ARG,
               "filter_columns_fill: out_nnz is %lld but the workspace holds a count of %lld", (long long)out_nnz,
               (long long)total);
  int64_t blocks = ceil_div(n_words, kBlock / 64);
  if (blocks > 256 * 32) blocks = 256 * 32;
  hipLaunchKernelGGL(filter_fill_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, st, a->nnz, a->col_idx, a->val,
                     words, before, out_col, out_val, out_src);
  GCNX_CHECK_LAUNCH();
  return GCNX_OK;
}

extern "C" int gcnx_scatter_rows_f32(int64_t n_idx, const int32_t *idx, const float *src,
                                     int64_t lds, int32_t width, float *dst, int64_t ldd,
                                     void *stream) {
  GCNX_REQUIRE(n_idx >= 0 && width >= 0, GCNX_ERR_INVALID_ARG, "scatter_rows: negative size");
  if (n_idx == 0 || width == 0) return GCNX_OK;
  GCNX_REQUIRE(idx && src && dst && lds >= width && ldd >= width, GCNX_ERR_INVALID_ARG,
               "scatter_rows: bad argument");
  const int vec = (width % 4 == 0) && (lds % 4 == 0) && (ldd % 4 == 0) &&
                  ((reinterpret_cast<uintptr_t>(src) & 15) == 0) &&
                  ((reinterpret_cast<uintptr_t>(dst) & 15) == 0);
  const int64_t total = n_idx * (int64_t)(vec ? width / 4 : width);
  int64_t blocks = ceil_div(total, kBlock);
  if (blocks > 256 * 32) blocks = 256 * 32;
  hipLaunchKernelGGL(scatter_rows_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, as_stream(stream),
                     n_idx, idx, src, lds, width, dst, ldd, vec);
  GCNX_CHECK_LAUNCH();
  return GCNX_OK;
}

extern "C" size_t gcnx_csr_from_coo_workspace_bytes(int64_t n_rows, int64_t nnz) {
  if (n_rows < 0 || nnz < 0) return 0;
  return pair_layout(nnz, 32 + bits_for(n_rows)).total;
}

extern "C" int gcnx_csr_from_coo(int64_t n_rows, int64_t n_cols, int64_t nnz, const int32_t *row,
                                 const int32_t *col, int64_t coord_stride, const void *val,
                                 int val_is_f64, int64_t *row_ptr, int32_t *col_idx,
                                 float *val_out, int64_t *perm_out, void *ws, size_t ws_bytes,
                                 void *stream) {
  GCNX_REQUIRE(n_rows >= 0 && n_cols >= 0 && nnz >= 0 && coord_stride >= 1, GCNX_ERR_INVALID_ARG,
               "csr_from_coo: bad size");
  GCNX_REQUIRE(n_rows < INT32_MAX && n_cols < INT32_MAX, GCNX_ERR_UNSUPPORTED,
               "csr_from_coo: more than 2^31-1 rows or columns");
  GCNX_REQUIRE(row_ptr, GCNX_ERR_INVALID_ARG, "csr_from_coo: null row_ptr");
  GCNX_REQUIRE(nnz == 0 || (row && col && val && col_idx && val_out), GCNX_ERR_INVALID_ARG,
               "csr_from_coo: null array");
  const int end_bit = 32 + bits_for(n_rows);
  const PairLayout l = pair_layout(nnz, end_bit);
  GCNX_REQUIRE(ws && ws_bytes >= l.total, GCNX_ERR_WORKSPACE,
               "csr_from_coo: workspace of %zu bytes required, got %zu", l.total, ws_bytes);
  hipStream_t st = as_stream(stream);
  char *w = static_cast<char *>(ws);
  int *bad = reinterpret_cast<int *>(w + l.flag);
  GCNX_CHECK_HIP(hipMemsetAsync(bad, 0, sizeof(int), st));
  if (nnz > 0) {
    hipLaunchKernelGGL(coo_keys_kernel, dim3((unsigned)ceil_div(nnz, 256)), dim3(256), 0, st, nnz,
                       n_rows, n_cols, row, col, coord_stride,
                       reinterpret_cast<uint64_t *>(w + l.keys_in),
                       reinterpret_cast<int64_t *>(w + l.idx_in), bad);
    GCNX_CHECK_LAUNCH();
  }
  int h_bad = 0;
  GCNX_CHECK_HIP(hipMemcpyAsync(&h_bad, bad, sizeof(int), hipMemcpyDeviceToHost, st));
  GCNX_CHECK_HIP(hipStreamSynchronize(st));
  GCNX_REQUIRE(h_bad == 0, GCNX_ERR_INDEX_RANGE,
               "csr_from_coo: a coordinate lies outside %lld x %lld", (long long)n_rows,
               (long long)n_cols);
  int rc = finish_from_pairs(l, w, n_rows, nnz, end_bit, val, val_is_f64, row_ptr, col_idx,
                             val_out, perm_out, st);
  if (rc != GCNX_OK) return rc;
  GCNX_CHECK_HIP(hipStreamSynchronize(st));
  return GCNX_OK;
}

extern "C" size_t gcnx_csr_transpose_workspace_bytes(int64_t n_rows, int64_t n_cols, int64_t nnz) {
  if (n_rows < 0 || n_cols < 0 || nnz < 0) return 0;
  return pair_layout(nnz, 32 + bits_for(n_cols)).total;
}

extern "C" int gcnx_csr_transpose(const gcnx_csr *a, int64_t *t_row_ptr, int32_t *t_col_idx,
                                  float *t_val, int64_t *perm_out, void *ws, size_t ws_bytes,
                                  void *stream) {
  GCNX_REQUIRE(a && t_row_ptr, GCNX_ERR_INVALID_ARG, "csr_transpose: null argument");
  GCNX_REQUIRE(a->n_rows >= 0 && a->n_cols >= 0 && a->nnz >= 0, GCNX_ERR_INVALID_ARG,
               "csr_transpose: bad size");
  GCNX_REQUIRE(a->n_rows < INT32_MAX && a->n_cols < INT32_MAX, GCNX_ERR_UNSUPPORTED,
               "csr_transpose: more than 2^31-1 rows or columns");
  GCNX_REQUIRE(a->nnz == 0 || (a->row_ptr && a->col_idx && a->val && t_col_idx && t_val),
               GCNX_ERR_INVALID_ARG, "csr_transpose: null array");
  const int end_bit = 32 + bits_for(a->n_cols);
  const PairLayout l = pair_layout(a->nnz, end_bit);
  GCNX_REQUIRE(ws && ws_bytes >= l.total, GCNX_ERR_WORKSPACE,
               "csr_transpose: workspace of %zu bytes required, got %zu", l.total, ws_bytes);
  hipStream_t st = as_stream(stream);
  char *w = static_cast<char *>(ws);
  if (a->nnz > 0) {
    hipLaunchKernelGGL(transpose_keys_kernel, dim3((unsigned)ceil_div(a->nnz, 256)), dim3(256), 0,
                       st, a->nnz, a->n_rows, a->row_ptr, a->col_idx,
                       reinterpret_cast<uint64_t *>(w + l.keys_in),
                       reinterpret_cast<int64_t *>(w + l.idx_in));
    GCNX_CHECK_LAUNCH();
  }
  int rc = finish_from_pairs(l, w, a->n_cols, a->nnz, end_bit, a->val, 0, t_row_ptr, t_col_idx,
                             t_val, perm_out, st);
  if (rc != GCNX_OK) return rc;
  GCNX_CHECK_HIP(hipStreamSynchronize(st));
  return GCNX_OK;
}

extern "C" size_t gcnx_csr_from_keys_workspace_bytes(int64_t n_keys, int64_t n_local_rows) {
  if (n_keys < 0 || n_local_rows < 0) return 0;
  return key_layout(n_keys + n_local_rows).total;
}

extern "C" int gcnx_csr_from_keys(uint64_t *keys, int64_t n_keys, int64_t row_begin,
                                  int64_t row_end, int add_identity, int64_t *row_ptr,
                                  int32_t *col_idx, float *val, int64_t *h_nnz, void *ws,
                                  size_t ws_bytes, void *stream) {
  GCNX_REQUIRE(keys && row_ptr && col_idx && val && h_nnz, GCNX_ERR_INVALID_ARG,
               "csr_from_keys: null argument");
  GCNX_REQUIRE(n_keys >= 0 && row_begin >= 0 && row_end >= row_begin && row_end < INT32_MAX,
               GCNX_ERR_INVALID_ARG, "csr_from_keys: bad size");
  const int64_t n_local = row_end - row_begin;
  const int64_t total = n_keys + (add_identity ? n_local : 0);
  const KeyLayout l = key_layout(n_keys + n_local);
  GCNX_REQUIRE(ws && ws_bytes >= l.total, GCNX_ERR_WORKSPACE,
               "csr_from_keys: workspace of %zu bytes required, got %zu", l.total, ws_bytes);
  hipStream_t st = as_stream(stream);
  char *w = static_cast<char *>(ws);
  uint64_t *sorted = reinterpret_cast<uint64_t *>(w + l.keys_sorted);
  int64_t *pos = reinterpret_cast<int64_t *>(w + l.pos);
  *h_nnz = 0;
  if (add_identity && n_local > 0) {
    // caller reserved n_local extra key slots behind the n_keys edges
    hipLaunchKernelGGL(append_diag_kernel, dim3((unsigned)ceil_div(n_local, 256)), dim3(256), 0, st,
                       row_begin, n_local, keys + n_keys);
    GCNX_CHECK_LAUNCH();
  }
  int64_t n_unique = 0;
  if (total > 0) {
    size_t tb = l.temp_bytes;
    GCNX_CHECK_HIP(rocprim::radix_sort_keys(w + l.temp, tb, keys, sorted, (size_t)total, 0u, 64u, st));
    const unsigned g = (unsigned)ceil_div(total, 256);
    hipLaunchKernelGGL(unique_flag_kernel, dim3(g), dim3(256), 0, st, total, sorted, pos);
    GCNX_CHECK_LAUNCH();
    // last flag + last exclusive position = number of unique keys
    int64_t last_flag = 0, last_pos = 0;
    GCNX_CHECK_HIP(hipMemcpyAsync(&last_flag, pos + total - 1, 8, hipMemcpyDeviceToHost, st));
    tb = l.temp_bytes;
    GCNX_CHECK_HIP(rocprim::exclusive_scan(w + l.temp, tb, pos, pos, (int64_t)0, (size_t)total,
                                           rocprim::plus<int64_t>(), st));
    GCNX_CHECK_HIP(hipMemcpyAsync(&last_pos, pos + total - 1, 8, hipMemcpyDeviceToHost, st));
    // unique keys go back into the caller's key buffer (it is scratch by now)
    hipLaunchKernelGGL(unique_scatter_kernel, dim3(g), dim3(256), 0, st, total, sorted, pos, keys);
    GCNX_CHECK_LAUNCH();
    GCNX_CHECK_HIP(hipStreamSynchronize(st));
    n_unique = last_pos + last_flag;
    hipLaunchKernelGGL(keys_to_csr_kernel, dim3((unsigned)ceil_div(n_unique, 256)), dim3(256), 0, st,
                       n_unique, keys, col_idx, val);
    GCNX_CHECK_LAUNCH();
  }
  hipLaunchKernelGGL(row_ptr_kernel, dim3((unsigned)ceil_div(n_local + 1, 256)), dim3(256), 0, st,
                     n_local, row_begin, n_unique, keys, row_ptr);
  GCNX_CHECK_LAUNCH();
  GCNX_CHECK_HIP(hipStreamSynchronize(st));
  *h_nnz = n_unique;
  return GCNX_OK;
}
