// Glue kernels of one training step: masked softmax cross-entropy (+accuracy,
// +gradient), dense dropout, relu/dropout backward, Adam, Bernoulli masks,
// degree normalisation, support interleaving.  All are streaming or tiny; the
// reductions are two-level with a fixed order (no float atomics), so results are
// reproducible run to run.
#include "gcnx_common.h"
#include "gcnx_philox.h"

namespace gcnx {

typedef float __attribute__((ext_vector_type(4))) f4;
typedef unsigned char __attribute__((ext_vector_type(4))) u8x4;

// ---------------------------------------------------------------- reductions
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// Sum of `v` over the block; valid in thread 0.  `sh` has kWavesPerBlock slots.
__device__ __forceinline__ double block_sum(double v, double *sh) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[w] = v;
  __syncthreads();
  double t = 0.0;
  if (threadIdx.x == 0)
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += sh[i];
  return t;
}

// ws layout (doubles): [0] = sum(mask); [1..nb] count partials; then loss
// partials [nb], hit partials [nb].
__global__ __launch_bounds__(kBlock) void xent_count_kernel(int64_t n, const uint8_t *__restrict__ mask,
                                                           double *__restrict__ part) {
  __shared__ double sh[kWavesPerBlock];
  double c = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    c += mask[i] ? 1.0 : 0.0;
  c = block_sum(c, sh);
  if (threadIdx.x == 0) part[blockIdx.x] = c;
}

__global__ __launch_bounds__(kBlock) void final_sum_kernel(int nb, const double *__restrict__ part,
                                                          double *__restrict__ out) {
  __shared__ double sh[kWavesPerBlock];
  double c = 0.0;
  for (int i = threadIdx.x; i < nb; i += blockDim.x) c += part[i];
  c = block_sum(c, sh);
  if (threadIdx.x == 0) *out = c;
}

__global__ __launch_bounds__(kBlock) void xent_rows_kernel(
    int64_t n, int32_t nc, const float *__restrict__ logits, int64_t ld,
    const float *__restrict__ labels, int64_t ldl, const uint8_t *__restrict__ mask,
    const double *__restrict__ msum_p, double msum_known, float *__restrict__ grad, int64_t ldg,
    double *__restrict__ loss_part, double *__restrict__ hit_part) {
  __shared__ double sh[kWavesPerBlock];
  const double msum = msum_known > 0.0 ? msum_known : *msum_p;   // a total given by the caller needs no counting pass
  const float wgt = msum > 0.0 ? (float)(1.0 / msum) : 0.0f;
  double loss = 0.0, hit = 0.0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const float *z = logits + i * ld;
    const float *y = labels + i * ldl;
    const bool on = mask[i] != 0;
    float zmax = z[0];
    int az = 0, ay = 0;
    float ybest = y[0];
    for (int j = 1; j < nc; ++j) {
      const float zj = z[j], yj = y[j];
      if (zj > zmax) { zmax = zj; az = j; }
      if (yj > ybest) { ybest = yj; ay = j; }
    }
    float se = 0.0f, ysum = 0.0f;
    for (int j = 0; j < nc; ++j) { se += __expf(z[j] - zmax); ysum += y[j]; }
    const float lse = __logf(se);
    if (on) {
      float ce = 0.0f;
      for (int j = 0; j < nc; ++j) ce -= y[j] * ((z[j] - zmax) - lse);
      loss += (double)ce;
      hit += (az == ay) ? 1.0 : 0.0;
    }
    if (grad) {
      float *g = grad + i * ldg;
      const float w = on ? wgt : 0.0f;
      const float inv = 1.0f / se;
      for (int j = 0; j < nc; ++j) g[j] = (__expf(z[j] - zmax) * inv * ysum - y[j]) * w;
    }
  }
  loss = block_sum(loss, sh);
  if (threadIdx.x == 0) loss_part[blockIdx.x] = loss;
  hit = block_sum(hit, sh);
  if (threadIdx.x == 0) hit_part[blockIdx.x] = hit;
}

// Same computation with a row spread over G = pow2 >= n_class lanes (G <= 64):
// lane j of a group holds class j, so logits / labels / grad are read and written
// fully coalesced and the row maxima, sums and first-maximum indices are formed
// with wavefront shuffles inside the group.
// Partner value for step `off` of a butterfly reduction inside a group of lanes.
// Up to 16 lanes the exchange is a DPP move on the vector ALU (no LDS crossbar):
// quad permutes for 1 and 2; for 4 and 8 the mirror of 8 / 16 lanes, which pairs
// the same sub-groups as xor does once every lane of a sub-group holds the
// sub-group's result (true after the earlier steps of the reduction).
template <int OFF>
__device__ __forceinline__ int butterfly_i(int v) {
  if constexpr (OFF == 1) return __builtin_amdgcn_mov_dpp(v, 0xB1, 0xF, 0xF, true);
  else if constexpr (OFF == 2) return __builtin_amdgcn_mov_dpp(v, 0x4E, 0xF, 0xF, true);
  else if constexpr (OFF == 4) return __builtin_amdgcn_mov_dpp(v, 0x141, 0xF, 0xF, true);
  else if constexpr (OFF == 8) return __builtin_amdgcn_mov_dpp(v, 0x140, 0xF, 0xF, true);
  else return __shfl_xor(v, OFF, kWave);
}
template <int OFF>
__device__ __forceinline__ float butterfly_f(float v) {
  return __int_as_float(butterfly_i<OFF>(__float_as_int(v)));
}
// One reduction step for the four running values of a row.
template <int OFF, int G>
__device__ __forceinline__ void xent_step_max(float &zmax, float &ymax) {
  if constexpr (OFF < G) {
    zmax = fmaxf(zmax, butterfly_f<OFF>(zmax));
    ymax = fmaxf(ymax, butterfly_f<OFF>(ymax));
    xent_step_max<OFF * 2, G>(zmax, ymax);
  }
}
template <int OFF, int G>
__device__ __forceinline__ void xent_step_sum(int &az, int &ay, float &se, float &ysum) {
  if constexpr (OFF < G) {
    az = min(az, butterfly_i<OFF>(az));
    ay = min(ay, butterfly_i<OFF>(ay));
    se += butterfly_f<OFF>(se);
    ysum += butterfly_f<OFF>(ysum);
    xent_step_sum<OFF * 2, G>(az, ay, se, ysum);
  }
}

template <int G>
__global__ __launch_bounds__(kBlock) void xent_rows_group_kernel(
    int64_t n, int32_t nc, const float *__restrict__ logits, int64_t ld,
    const float *__restrict__ labels, int64_t ldl, const uint8_t *__restrict__ mask,
    const double *__restrict__ msum_p, double msum_known, float *__restrict__ grad, int64_t ldg,
    double *__restrict__ loss_part, double *__restrict__ hit_part) {
  __shared__ double sh[kWavesPerBlock];
  constexpr int RPB = kBlock / G;  // rows per block and pass
  const double msum = msum_known > 0.0 ? msum_known : *msum_p;
  const float wgt = msum > 0.0 ? (float)(1.0 / msum) : 0.0f;
  const int sub = threadIdx.x % G;
  const bool in = sub < nc;
  double loss = 0.0, hit = 0.0;
  for (int64_t base = (int64_t)blockIdx.x * RPB; base < n; base += (int64_t)gridDim.x * RPB) {
    const int64_t row = base + threadIdx.x / G;
    const bool valid = row < n;
    const float ninf = -__builtin_inff();
    const float z = (valid && in) ? logits[row * ld + sub] : ninf;
    const float y = (valid && in) ? labels[row * ldl + sub] : ninf;
    float zmax = z, ymax = y;
    xent_step_max<1, G>(zmax, ymax);
    int az = (in && z == zmax) ? sub : G, ay = (in && y == ymax) ? sub : G;  // first maximum wins
    const float e = (valid && in) ? __expf(z - zmax) : 0.0f;
    float se = e, ysum = (valid && in) ? y : 0.0f;
    xent_step_sum<1, G>(az, ay, se, ysum);
    if (valid) {
      const bool on = mask[row] != 0;
      if (on && in) loss += (double)(-y * ((z - zmax) - __logf(se)));
      if (on && sub == 0 && az == ay) hit += 1.0;
      if (grad && in) grad[row * ldg + sub] = (e / se * ysum - y) * (on ? wgt : 0.0f);
    }
  }
  loss = block_sum(loss, sh);
  if (threadIdx.x == 0) loss_part[blockIdx.x] = loss;
  hit = block_sum(hit, sh);
  if (threadIdx.x == 0) hit_part[blockIdx.x] = hit;
}

// The same with four consecutive classes per lane (16-byte loads and stores:
// a wavefront instruction moves 1 KiB instead of 256 bytes): L = pow2 >= nc / 4
// lanes per row, nc a multiple of 4, rows 16-byte aligned.
template <int L>
__global__ __launch_bounds__(kBlock) void xent_rows_vec4_kernel(
    int64_t n, int32_t nc, const float *__restrict__ logits, int64_t ld,
    const float *__restrict__ labels, int64_t ldl, const uint8_t *__restrict__ mask,
    const double *__restrict__ msum_p, double msum_known, float *__restrict__ grad, int64_t ldg,
    double *__restrict__ loss_part, double *__restrict__ hit_part) {
  typedef float __attribute__((ext_vector_type(4))) f4;
  __shared__ double sh[kWavesPerBlock];
  constexpr int RPB = kBlock / L;
  const double msum = msum_known > 0.0 ? msum_known : *msum_p;
  const float wgt = msum > 0.0 ? (float)(1.0 / msum) : 0.0f;
  const int sub = threadIdx.x % L;
  const int c0 = 4 * sub;
  const bool in = c0 < nc;
  const float ninf = -__builtin_inff();
  double loss = 0.0, hit = 0.0;
  for (int64_t base = (int64_t)blockIdx.x * RPB; base < n; base += (int64_t)gridDim.x * RPB) {
    const int64_t row = base + threadIdx.x / L;
    const bool valid = row < n;
    f4 z = f4(ninf), y = f4(ninf);
    if (valid && in) {
      z = *reinterpret_cast<const f4 *>(logits + row * ld + c0);
      y = *reinterpret_cast<const f4 *>(labels + row * ldl + c0);
    }
    float zmax = fmaxf(fmaxf(z[0], z[1]), fmaxf(z[2], z[3]));
    float ymax = fmaxf(fmaxf(y[0], y[1]), fmaxf(y[2], y[3]));
    xent_step_max<1, L>(zmax, ymax);
    int az = 4 * L, ay = 4 * L;                       // first maximum wins
#pragma unroll
    for (int j = 3; j >= 0; --j) {
      if (in && z[j] == zmax) az = c0 + j;
      if (in && y[j] == ymax) ay = c0 + j;
    }
    f4 e = f4(0.0f);
    float se = 0.0f, ysum = 0.0f;
    if (valid && in) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        e[j] = __expf(z[j] - zmax);
        se += e[j];
        ysum += y[j];
      }
    }
    xent_step_sum<1, L>(az, ay, se, ysum);
    if (valid) {
      const bool on = mask[row] != 0;
      if (on && in) {
        const float lse = __logf(se);
        float t = 0.0f;
#pragma unroll
        for (int j = 0; j < 4; ++j) t += -y[j] * ((z[j] - zmax) - lse);
        loss += (double)t;
      }
      if (on && sub == 0 && az == ay) hit += 1.0;
      if (grad && in) {
        const float w = on ? wgt : 0.0f, inv = ysum / se;
        f4 gq;
#pragma unroll
        for (int j = 0; j < 4; ++j) gq[j] = (e[j] * inv - y[j]) * w;
        *reinterpret_cast<f4 *>(grad + row * ldg + c0) = gq;
      }
    }
  }
  loss = block_sum(loss, sh);
  if (threadIdx.x == 0) loss_part[blockIdx.x] = loss;
  hit = block_sum(hit, sh);
  if (threadIdx.x == 0) hit_part[blockIdx.x] = hit;
}

__global__ __launch_bounds__(kBlock) void xent_final_kernel(int nb, const double *__restrict__ ws_msum,
                                                           double msum_known,
                                                           const double *__restrict__ loss_part,
                                                           const double *__restrict__ hit_part,
                                                           float *__restrict__ out) {
  __shared__ double sh[kWavesPerBlock];
  double l = 0.0, h = 0.0;
  for (int i = threadIdx.x; i < nb; i += blockDim.x) { l += loss_part[i]; h += hit_part[i]; }
  l = block_sum(l, sh);
  h = block_sum(h, sh);
  if (threadIdx.x == 0) {
    const double m = msum_known > 0.0 ? msum_known : *ws_msum;
    out[0] = m > 0.0 ? (float)(l / m) : 0.0f;
    out[1] = m > 0.0 ? (float)(h / m) : 0.0f;
  }
}

static int xent_blocks(int64_t n) {
  int64_t b = ceil_div(n, kBlock);
  return (int)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

// ---------------------------------------------------------------- elementwise
__global__ __launch_bounds__(kBlock) void dropout_dense_kernel(int64_t n, const float *__restrict__ x,
                                                              const uint8_t *__restrict__ mask,
                                                              float scale, float *__restrict__ out,
                                                              int vec_ok) {
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t n4 = vec_ok ? n / 4 : 0;
  for (int64_t i = tid; i < n4; i += stride) {
    const f4 v = reinterpret_cast<const f4 *>(x)[i];
    const u8x4 m = reinterpret_cast<const u8x4 *>(mask)[i];
    f4 o;
    o.x = m.x ? v.x * scale : 0.0f;
    o.y = m.y ? v.y * scale : 0.0f;
    o.z = m.z ? v.z * scale : 0.0f;
    o.w = m.w ? v.w * scale : 0.0f;
    reinterpret_cast<f4 *>(out)[i] = o;
  }
  for (int64_t i = n4 * 4 + tid; i < n; i += stride) out[i] = mask[i] ? x[i] * scale : 0.0f;
}

__global__ __launch_bounds__(kBlock) void relu_dropout_bwd_kernel(
    int64_t n, const float *__restrict__ y, const float *__restrict__ dy,
    const uint8_t *__restrict__ mask, float scale, float *__restrict__ dx, int vec_ok) {
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t n4 = vec_ok ? n / 4 : 0;
  for (int64_t i = tid; i < n4; i += stride) {
    const f4 yy = reinterpret_cast<const f4 *>(y)[i];
    f4 g = reinterpret_cast<const f4 *>(dy)[i];
    if (mask) {
      const u8x4 m = reinterpret_cast<const u8x4 *>(mask)[i];
      g.x = m.x ? g.x * scale : 0.0f;
      g.y = m.y ? g.y * scale : 0.0f;
      g.z = m.z ? g.z * scale : 0.0f;
      g.w = m.w ? g.w * scale : 0.0f;
    }
    f4 o;
    o.x = yy.x > 0.0f ? g.x : 0.0f;
    o.y = yy.y > 0.0f ? g.y : 0.0f;
    o.z = yy.z > 0.0f ? g.z : 0.0f;
    o.w = yy.w > 0.0f ? g.w : 0.0f;
    reinterpret_cast<f4 *>(dx)[i] = o;
  }
  for (int64_t i = n4 * 4 + tid; i < n; i += stride) {
    float g = dy[i];
    if (mask) g = mask[i] ? g * scale : 0.0f;
    dx[i] = y[i] > 0.0f ? g : 0.0f;
  }
}

// t_dev (nullable): the step count is read from the device and the bias-corrected
// rate lr * sqrt(1 - b2^t) / (1 - b1^t) formed here (lr_t then holds plain lr).
__global__ __launch_bounds__(kBlock) void adam_kernel(int64_t n, float *__restrict__ w,
                                                     const float *__restrict__ g,
                                                     float *__restrict__ m, float *__restrict__ v,
                                                     float lr_t, float b1, float b2, float eps,
                                                     float wd, const uint64_t *__restrict__ t_dev) {
  if (t_dev) {
    const double t = (double)*t_dev;
    lr_t = (float)((double)lr_t * sqrt(1.0 - pow((double)b2, t)) / (1.0 - pow((double)b1, t)));
  }
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x) {
    const float wi = w[i];
    const float gi = g[i] + wd * wi;
    const float mi = m[i] + (gi - m[i]) * (1.0f - b1);
    const float vi = v[i] + (gi * gi - v[i]) * (1.0f - b2);
    m[i] = mi;
    v[i] = vi;
    w[i] = wi - lr_t * mi / (sqrtf(vi) + eps);
  }
}

// out[0] = (add ? add[0] : 0) + scale * sum_i w[i]^2 / 2.  One workgroup: the
// weight matrices this is for hold 10^4..10^5 elements, and a single fixed
// reduction tree keeps the sum reproducible.
constexpr int kL2Block = 1024;
__global__ __launch_bounds__(kL2Block) void l2_loss_kernel(int64_t n, const float *__restrict__ w, float scale,
                                                           const float *__restrict__ add,
                                                           float *__restrict__ out) {
  __shared__ float part[kL2Block / kWave];
  float s = 0.0f;
  for (int64_t i = threadIdx.x; i < n; i += kL2Block) s = fmaf(w[i], w[i], s);
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) s += __shfl_xor(s, off, kWave);
  if ((threadIdx.x & (kWave - 1)) == 0) part[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.0f;
#pragma unroll
    for (int k = 0; k < kL2Block / kWave; ++k) t += part[k];
    out[0] = (add ? add[0] : 0.0f) + scale * (0.5f * t);
  }
}

// One wavefront per 64 output words: word (i, w) collects the signs of
// x[row(i)][32w .. 32w+31] (bit j set <=> element > 0) with a wavefront ballot
// per 64 columns, so reads are coalesced and every lane writes at most once.
__global__ __launch_bounds__(kBlock) void pack_sign_bits_kernel(int64_t n_rows,
                                                               const int32_t *__restrict__ row_idx,
                                                               const float *__restrict__ x, int64_t ldx,
                                                               int32_t width, uint32_t *__restrict__ bits,
                                                               int64_t ldw) {
  const int lane = threadIdx.x & (kWave - 1);
  const int64_t wave = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)gridDim.x * kWavesPerBlock;
  for (int64_t i = wave; i < n_rows; i += nwaves) {
    const int64_t r = row_idx ? (int64_t)row_idx[i] : i;
    for (int c0 = 0; c0 < width; c0 += kWave) {
      const int c = c0 + lane;
      const bool on = c < width && x[r * ldx + c] > 0.0f;
      const uint64_t m = __ballot(on);
      if (lane == 0) bits[i * ldw + (c0 >> 5)] = (uint32_t)m;
      if (lane == 32 && c0 + 32 < width) bits[i * ldw + (c0 >> 5) + 1] = (uint32_t)(m >> 32);
    }
  }
}

// Dropout of gathered rows: out[i][c] = keep(offset + row_ids[i] * width + c) ?
// x[i][c] * scale : 0, i.e. row i draws the mask of row row_ids[i] of the matrix
// (row length `width`) the mask is defined on.  A thread handles 4 consecutive
// columns (one Philox block) when vec_ok, else one element.
__global__ __launch_bounds__(kBlock) void dropout_rows_philox_kernel(
    int64_t n_rows, int32_t width, const float *x, int64_t ldx,
    const int32_t *__restrict__ row_ids, float keep, float scale, uint64_t seed, uint64_t offset,
    float *out, int64_t ldo, int vec_ok) {   // out may be x (each element is read, then written, by one thread)
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  if (vec_ok) {
    const int w4 = width / 4;
    const int64_t total = n_rows * (int64_t)w4;
    for (int64_t t = tid; t < total; t += stride) {
      const int64_t i = t / w4;
      const int c = (int)(t % w4) * 4;
      const uint64_t e = offset + (uint64_t)row_ids[i] * (uint64_t)width + (uint64_t)c;
      const philox4 r = philox4x32_10(e >> 2, 0ull, seed);
      const f4 v = *reinterpret_cast<const f4 *>(x + i * ldx + c);
      f4 o;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        o[j] = ((float)(r.v[j] >> 8) * (1.0f / 16777216.0f) < keep) ? v[j] * scale : 0.0f;
      *reinterpret_cast<f4 *>(out + i * ldo + c) = o;
    }
  } else {
    const int64_t total = n_rows * (int64_t)width;
    for (int64_t t = tid; t < total; t += stride) {
      const int64_t i = t / width;
      const int c = (int)(t % width);
      const uint64_t e = offset + (uint64_t)row_ids[i] * (uint64_t)width + (uint64_t)c;
      const philox4 r = philox4x32_10(e >> 2, 0ull, seed);
      out[i * ldo + c] = ((float)(r.v[e & 3] >> 8) * (1.0f / 16777216.0f) < keep) ? x[i * ldx + c] * scale : 0.0f;
    }
  }
}

__global__ void counter_add_kernel(uint64_t *ctr, uint64_t inc) { *ctr += inc; }

// byte k of the mask = 1 iff uniform(seed, offset + k) < keep; four consecutive
// global positions share one Philox block.  step_dev (nullable) adds
// *step_dev * step_stride to the offset: the step counter then lives on the
// device, so a captured HIP graph draws a fresh mask on every replay.
__global__ __launch_bounds__(kBlock) void bernoulli_kernel(int64_t n, float keep, uint64_t seed,
                                                          uint64_t offset,
                                                          const uint64_t *__restrict__ step_dev,
                                                          uint64_t step_stride,
                                                          uint8_t *__restrict__ mask) {
  if (step_dev) offset += *step_dev * step_stride;
  const uint64_t first_blk = offset >> 2;
  const uint64_t last_blk = (offset + (uint64_t)n - 1) >> 2;
  const uint64_t nblk = last_blk - first_blk + 1;
  for (uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; b < nblk;
       b += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t blk = first_blk + b;
    const philox4 r = philox4x32_10(blk, 0ull, seed);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint64_t e = blk * 4 + j;
      if (e >= offset && e < offset + (uint64_t)n) {
        const float u = (float)(r.v[j] >> 8) * (1.0f / 16777216.0f);
        mask[e - offset] = u < keep ? 1 : 0;
      }
    }
  }
}

// Dropout with the mask regenerated from Philox instead of read from memory:
// element i is kept iff uniform(seed, offset + i) < keep (the same rule and bits
// as bernoulli_kernel).  MODE 0: out = keep_i ? x * scale : 0.  MODE 1 (backward
// through relu then dropout): out = (y > 0 && keep_i) ? dy * scale : 0, with
// x = dy and y the relu output.  Four consecutive global positions share one
// Philox block; with offset % 4 == 0 and aligned pointers a thread handles one
// block with 16-byte accesses.
template <int MODE>
__global__ __launch_bounds__(kBlock) void dropout_philox_kernel(
    int64_t n, const float *__restrict__ x, const float *__restrict__ y, float keep, float scale,
    uint64_t seed, uint64_t offset, float *__restrict__ out, uint8_t *__restrict__ mask_out,
    int vec_ok) {
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  if (vec_ok) {
    const int64_t n4 = n / 4;
    const uint64_t blk0 = offset >> 2;
    for (int64_t i = tid; i < n4; i += stride) {
      const philox4 r = philox4x32_10(blk0 + (uint64_t)i, 0ull, seed);
      const f4 v = reinterpret_cast<const f4 *>(x)[i];
      f4 yy = {1.0f, 1.0f, 1.0f, 1.0f};
      if (MODE == 1) yy = reinterpret_cast<const f4 *>(y)[i];
      f4 o;
      u8x4 m;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bool kp = (float)(r.v[j] >> 8) * (1.0f / 16777216.0f) < keep;
        m[j] = kp ? 1 : 0;
        o[j] = (kp && yy[j] > 0.0f) ? v[j] * scale : 0.0f;
      }
      reinterpret_cast<f4 *>(out)[i] = o;
      if (mask_out) reinterpret_cast<u8x4 *>(mask_out)[i] = m;
    }
    for (int64_t i = n4 * 4 + tid; i < n; i += stride) {
      const uint64_t e = offset + (uint64_t)i;
      const philox4 r = philox4x32_10(e >> 2, 0ull, seed);
      const bool kp = (float)(r.v[e & 3] >> 8) * (1.0f / 16777216.0f) < keep;
      const float yv = MODE == 1 ? y[i] : 1.0f;
      out[i] = (kp && yv > 0.0f) ? x[i] * scale : 0.0f;
      if (mask_out) mask_out[i] = kp ? 1 : 0;
    }
  } else {
    for (int64_t i = tid; i < n; i += stride) {
      const uint64_t e = offset + (uint64_t)i;
      const philox4 r = philox4x32_10(e >> 2, 0ull, seed);
      const bool kp = (float)(r.v[e & 3] >> 8) * (1.0f / 16777216.0f) < keep;
      const float yv = MODE == 1 ? y[i] : 1.0f;
      out[i] = (kp && yv > 0.0f) ? x[i] * scale : 0.0f;
      if (mask_out) mask_out[i] = kp ? 1 : 0;
    }
  }
}

// out[i] = lo + (hi - lo) * uniform(seed, offset + i), uniform = (word >> 8) * 2^-24
__global__ __launch_bounds__(kBlock) void philox_uniform_kernel(int64_t n, float lo, float hi,
                                                               uint64_t seed, uint64_t offset,
                                                               float *__restrict__ out) {
  const uint64_t first_blk = offset >> 2;
  const uint64_t last_blk = (offset + (uint64_t)n - 1) >> 2;
  const uint64_t nblk = last_blk - first_blk + 1;
  for (uint64_t b = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; b < nblk;
       b += (uint64_t)gridDim.x * blockDim.x) {
    const uint64_t blk = first_blk + b;
    const philox4 r = philox4x32_10(blk, 0ull, seed);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const uint64_t e = blk * 4 + j;
      if (e >= offset && e < offset + (uint64_t)n)
        out[e - offset] = lo + (hi - lo) * ((float)(r.v[j] >> 8) * (1.0f / 16777216.0f));
    }
  }
}

__device__ __forceinline__ int64_t row_of_slot2(const int64_t *row_ptr, int64_t n_rows, int64_t s) {
  int64_t lo = 0, hi = n_rows;
  while (hi - lo > 1) {
    const int64_t mid = lo + (hi - lo) / 2;
    if (row_ptr[mid] <= s) lo = mid; else hi = mid;
  }
  return lo;
}

__global__ __launch_bounds__(kBlock) void normalize_kernel(
    int64_t nnz, int64_t n_rows, int64_t row_begin, const int64_t *__restrict__ row_ptr,
    const int32_t *__restrict__ col_idx, const float *__restrict__ val,
    const int32_t *__restrict__ row_sum, const double *__restrict__ d_table, int64_t table_len,
    float *__restrict__ out, int *__restrict__ bad) {
  const int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= nnz) return;
  const int64_t r = row_of_slot2(row_ptr, n_rows, s) + row_begin;
  const int64_t c = col_idx[s];
  const int64_t kr = row_sum[r], kc = row_sum[c];
  if (kr < 0 || kr >= table_len || kc < 0 || kc >= table_len) {
    atomicOr(bad, 1);
    out[s] = 0.0f;
    return;
  }
  // value at (r,c) of (A.D)^T.D for a symmetric A: fl(fl(a * d[r]) * d[c]);
  // __dmul_rn keeps the two roundings separate (no contraction).
  const double t = __dmul_rn((double)val[s], d_table[kr]);
  out[s] = (float)__dmul_rn(t, d_table[kc]);
}

__global__ __launch_bounds__(kBlock) void interleave_kernel(
    int64_t nnz, int32_t n_sup, int32_t i, const int32_t *__restrict__ row,
    const int32_t *__restrict__ col, int64_t stride, const void *__restrict__ val, int is_f64,
    int32_t *__restrict__ out_row, int32_t *__restrict__ out_col, void *__restrict__ out_val) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= nnz) return;
  out_row[k] = row[k * stride];
  out_col[k] = col[k * stride] * n_sup + i;
  if (is_f64)
    static_cast<double *>(out_val)[k] = static_cast<const double *>(val)[k];
  else
    static_cast<float *>(out_val)[k] = static_cast<const float *>(val)[k];
}

static unsigned stream_grid(int64_t n_items) {
  int64_t b = ceil_div(n_items, kBlock);
  if (b < 1) b = 1;
  if (b > 256 * 16) b = 256 * 16;
  return (unsigned)b;
}

static inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
static inline bool aligned4(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 3) == 0; }

}  // namespace gcnx

using namespace gcnx;

extern "C" size_t gcnx_xent_workspace_bytes(int64_t n) {
  return (size_t)(1 + 3 * (size_t)xent_blocks(n < 0 ? 0 : n)) * sizeof(double);
}

extern "C" int gcnx_masked_softmax_xent_f32(int64_t n, int32_t n_class, const float *logits,
                                            int64_t ld, const float *labels, int64_t ldl,
                                            const uint8_t *mask, double mask_total,
                                            float *out_loss_acc, float *grad, int64_t ldg,
                                            void *ws, size_t ws_bytes, void *stream) {
  GCNX_REQUIRE(n > 0 && n_class > 0 && logits && labels && mask && out_loss_acc,
               GCNX_ERR_INVALID_ARG, "xent: bad argument");
  GCNX_REQUIRE(ld >= n_class && ldl >= n_class && (!grad || ldg >= n_class), GCNX_ERR_INVALID_ARG,
               "xent: leading dimension smaller than n_class");
  GCNX_REQUIRE(ws && ws_bytes >= gcnx_xent_workspace_bytes(n), GCNX_ERR_WORKSPACE,
               "xent: workspace of %zu bytes required", gcnx_xent_workspace_bytes(n));
  hipStream_t st = as_stream(stream);
  const int nb = xent_blocks(n);
  double *w = static_cast<double *>(ws);
  double *msum = w, *cnt = w + 1, *lp = w + 1 + nb, *hp = w + 1 + 2 * nb;
  const double known = mask_total > 0.0 ? mask_total : 0.0;
  if (known == 0.0) {
    hipLaunchKernelGGL(xent_count_kernel, dim3(nb), dim3(kBlock), 0, st, n, mask, cnt);
    hipLaunchKernelGGL(final_sum_kernel, dim3(1), dim3(kBlock), 0, st, nb, cnt, msum);
  }
#define GCNX_XENT(G)                                                                          \
  hipLaunchKernelGGL((xent_rows_group_kernel<G>), dim3(nb), dim3(kBlock), 0, st, n, n_class, logits, \
                     ld, labels, ldl, mask, msum, known, grad, ldg, lp, hp)
  const bool vec4 = n_class % 4 == 0 && n_class <= 64 && ld % 4 == 0 && ldl % 4 == 0 && aligned16(logits) &&
                    aligned16(labels) && (!grad || (ldg % 4 == 0 && aligned16(grad)));
#define GCNX_XENT4(L)                                                                          \
  hipLaunchKernelGGL((xent_rows_vec4_kernel<L>), dim3(nb), dim3(kBlock), 0, st, n, n_class, logits, \
                     ld, labels, ldl, mask, msum, known, grad, ldg, lp, hp)
  if (vec4) {
    if (n_class > 32) GCNX_XENT4(16);
    else if (n_class > 16) GCNX_XENT4(8);
    else if (n_class > 8) GCNX_XENT4(4);
    else if (n_class > 4) GCNX_XENT4(2);
    else GCNX_XENT4(1);
  } else if (n_class > 64) {
    hipLaunchKernelGGL(xent_rows_kernel, dim3(nb), dim3(kBlock), 0, st, n, n_class, logits, ld, labels,
                       ldl, mask, msum, known, grad, ldg, lp, hp);
  } else if (n_class > 32) GCNX_XENT(64);
  else if (n_class > 16) GCNX_XENT(32);
  else if (n_class > 8) GCNX_XENT(16);
  else if (n_class > 4) GCNX_XENT(8);
  else if (n_class > 2) GCNX_XENT(4);
  else if (n_class > 1) GCNX_XENT(2);
  else GCNX_XENT(1);
#undef GCNX_XENT
#undef GCNX_XENT4
  hipLaunchKernelGGL(xent_final_kernel, dim3(1), dim3(kBlock), 0, st, nb, msum, known, lp, hp, out_loss_acc);
  GCNX_CHECK_LAUNCH();
  return GCNX_OK;
}

extern "C" int gcnx_dropout_dense_f32(int64_t n, const float *x, const uint8_t *mask, float scale,
                                      float *out, void *stream) {
  GCNX_REQUIRE(n >= 0 && (n == 0 || (x && mask && out)), GCNX_ERR_INVALID_ARG, "dropout: bad argument");
  if (n == 0) return GCNX_OK;
  const int vec_ok = aligned16(x) && aligned16(out) && aligned4(mask);
  hipLaunchKernelGGL(dropout_dense_kernel, dim3(stream_grid(n / 4 + 1)), dim3(kBlock), 0,
                     as_stream(stream), n, x, mask, scale, out, vec_ok);
  GCNX_CHECK_LAUNCH();
  return GCNX_OK;
}

extern "C" int gcnx_relu_dropout_bwd_f32(int64_t n, const float *y, const float *dy,
                                         const uint8_t *mask, float scale, float *dx, void *stream) {
  GCNX_REQUIRE(n >= 0 && (n == 0 || (y && dy && dx)), GCNX_ERR_INVALID_ARG, "relu_bwd: bad argument");
  if (n == 0) return GCNX_OK;
  const int vec_ok = aligned16(y) && aligned16(dy) && aligned16(dx) && (!mask || aligned4(mask));
  hipLaunchKernelGGL(relu_dropout_bwd_kernel, dim3(stream_grid(n / 4 + 1)), dim3(kBlock), 0,
                     as_stream(stream), n, y, dy, mask, scale, dx, vec_ok);
  GCNX_CHECK_LAUNCH();
  return GCNX_OK;
}

extern "C" int gcnx_adam_step_f32(int64_t n, float *w, const float *grad, float *m, float *v,
                                  float lr, float beta1, float beta2, float eps, float weight_decay,
                                  int64_t t, void *stream) {
  GCNX_REQUIRE(n >= 0 && t >= 1 && (n == 0 || (w && grad && m && v)), GCNX_ERR_INVALID_ARG,
               "adam: bad argument");
  if (n == 0) return GCNX_OK;
  const double lr_t = (double)lr * sqrt(1.0 - pow((double)beta2, (double)t)) /
                      (1.0 - pow((double)beta1, (double)t));
  hipLaunchKernelGGL(adam_kernel, dim3(stream_grid(n)), dim3(kBlock), 0, as_stream(stream), n, w,
                     grad, m, v, (float)lr_t, beta1, beta2, eps, weight_decay,
                     (const uint64_t *)nullptr);
  GCNX_CHECK_LAUNCH();
  return GCNX_OK;
}

extern "C" int gcnx_adam_step_dev_f32(int64_t n, float *w, const float *grad, float *m, float *v,
                                      float lr, float beta1, float beta2, float eps,
                                      float weight_decay, const uint64_t *t_dev, void *stream) {
  GCNX_REQUIRE(n >= 0 && t_dev && (n == 0 || (w && grad && m && v)), GCNX_ERR_INVALID_ARG,
               "adam_dev: bad argument");
  if (n == 0) return GCNX_OK;
  hipLaunchKernelGGL(adam_kernel, dim3(stream_grid(n)), dim3(kBlock), 0, as_stream(stream), n, w,
                     grad, m, v, lr, beta1, beta2, eps, weight_decay, t_dev);
  GCNX_CHECK_LAUNCH();
  return GCNX_OK;
}

extern "C" int gcnx_l2_loss_f32(int64_t n, const float *w, float scale, const float *add, float *out,
                                void *stream) {
  GCNX_REQUIRE(n >= 0 && out && (n == 0 || w), GCNX_ERR_INVALID_ARG, "l2_loss: bad argument");
  hipLaunchKernelGGL(l2_loss_kernel, dim3(1), dim3(kL2Block), 0, as_stream(stream), n, w, scale, add, out);
  GCNX_CHECK_LAUNCH();
  return GCNX_OK;
}

extern "C" int gcnx_pack_sign_bits_f32(int64_t n_rows, const int32_t *row_idx, const float *x,
                                       int64_t ldx, int32_t width, uint32_t *bits, int64_t ld_words,
                                       void *stream) {
  GCNX_REQUIRE(n_rows >= 0 && width >= 0 && ldx >= width && ld_words * 32 >= width,
               GCNX_ERR_INVALID_ARG, "pack_sign_bits: bad shape");
  if (n_rows == 0 || width == 0) return GCNX_OK;
  GCNX_REQUIRE(x && bits, GCNX_ERR_INVALID_ARG, "pack_sign_bits: null argument");
  int64_t blocks = ceil_div(n_rows, kWavesPerBlock);
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(pack_sign_bits_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, as_stream(stream),
                     n_rows, row_idx, x, ldx, width, bits, ld_words);
  GCNX_CHECK_LAUNCH();
  return GCNX_OK;
}

extern "C" int gcnx_dropout_rows_philox_f32(int64_t n_rows, int32_t width, const float *x, int64_t ldx,
                                            const int32_t *row_ids, float keep, uint64_t seed,
                                            uint64_t offset, float *out, int64_t ldo, void *stream) {
  GCNX_REQUIRE(n_rows >= 0 && width >= 0 && keep > 0.0f && ldx >= width && ldo >= width,
               GCNX_ERR_INVALID_ARG, "dropout_rows_philox: bad argument");
  if (n_rows == 0 || width == 0) return GCNX_OK;
  GCNX_REQUIRE(x && out && row_ids, GCNX_ERR_INVALID_ARG, "dropout_rows_philox: null argument");
  const int vec_ok = aligned16(x) && aligned16(out) && (offset % 4 == 0) && (width % 4 == 0) &&
                     (ldx % 4 == 0) && (ldo % 4 == 0);
  const int64_t work = vec_ok ? n_rows * (int64_t)(width / 4) : n_rows * (int64_t)width;
  hipLaunchKernelGGL(dropout_rows_philox_kernel, dim3(stream_grid(work)), dim3(kBlock), 0, as_stream(stream),
                     n_rows, width, x, ldx, row_ids, keep, 1.0f / keep, seed, offset, out, ldo, vec_ok);
  GCNX_CHECK_LAUNCH();
  return GCNX_OK;
}

extern "C" int gcnx_counter_add_u64(uint64_t *counter, uint64_t inc, void *stream) {
  GCNX_REQUIRE(counter, GCNX_ERR_INVALID_ARG, "counter_add: null counter");
  hipLaunchKernelGGL(counter_add_kernel, dim3(1), dim3(1), 0, as_stream(stream), counter, inc);
  GCNX_CHECK_LAUNCH();
  return GCNX_OK;
}

extern "C" int gcnx_bernoulli_mask_u8(int64_t n, float keep, uint64_t seed, uint64_t offset,
                                      uint8_t *mask, void *stream) {
  GCNX_REQUIRE(n >= 0 && (n == 0 || mask), GCNX_ERR_INVALID_ARG, "bernoulli: bad argument");
  if (n == 0) return GCNX_OK;
  hipLaunchKernelGGL(bernoulli_kernel, dim3(stream_grid(n / 4 + 2)), dim3(kBlock), 0,
                     as_stream(stream), n, keep, seed, offset, (const uint64_t *)nullptr,
                     (uint64_t)0, mask);
  GCNX_CHECK_LAUNCH();
  return GCNX_OK;
}

extern "C" int gcnx_bernoulli_mask_dev_u8(int64_t n, float keep, uint64_t seed, uint64_t offset,
                                          const uint64_t *step_dev, uint64_t step_stride,
                                          uint8_t *mask, void *stream) {
  GCNX_REQUIRE(n >= 0 && step_dev && (n == 0 || mask), GCNX_ERR_INVALID_ARG,
               "bernoulli_dev: bad argument");
  if (n == 0) return GCNX_OK;
  hipLaunchKernelGGL(bernoulli_kernel, dim3(stream_grid(n / 4 + 2)), dim3(kBlock), 0,
                     as_stream(stream), n, keep, seed, offset, step_dev, step_stride, mask);
  GCNX_CHECK_LAUNCH();
  return GCNX_OK;
}

extern "C" int gcnx_philox_uniform_f32(int64_t n, float lo, float hi, uint64_t seed,
                                       uint64_t offset, float *out, void *stream) {
  GCNX_REQUIRE(n >= 0 && (n == 0 || out), GCNX_ERR_INVALID_ARG, "philox_uniform: bad argument");
  if (n == 0) return GCNX_OK;
  hipLaunchKernelGGL(philox_uniform_kernel, dim3(stream_grid(n / 4 + 2)), dim3(kBlock), 0,
                     as_stream(stream), n, lo, hi, seed, offset, out);
  GCNX_CHECK_LAUNCH();
  return GCNX_OK;
}

extern "C" int gcnx_dropout_philox_f32(int64_t n, const float *x, float keep, uint64_t seed,
                                       uint64_t offset, float *out, uint8_t *mask_out,
                                       void *stream) {
  GCNX_REQUIRE(n >= 0 && keep > 0.0f && (n == 0 || (x && out)), GCNX_ERR_INVALID_ARG,
               "dropout_philox: bad argument");
  if (n == 0) return GCNX_OK;
  const int vec_ok = aligned16(x) && aligned16(out) && (!mask_out || aligned4(mask_out)) &&
                     (offset % 4 == 0);
  hipLaunchKernelGGL((dropout_philox_kernel<0>), dim3(stream_grid(n / 4 + 1)), dim3(kBlock), 0,
                     as_stream(stream), n, x, (const float *)nullptr, keep, 1.0f / keep, seed, offset,
                     out, mask_out, vec_ok);
  GCNX_CHECK_LAUNCH();
  return GCNX_OK;
}

extern "C" int gcnx_relu_dropout_bwd_philox_f32(int64_t n, const float *y, const float *dy,
                                                float keep, uint64_t seed, uint64_t offset,
                                                float *dx, void *stream) {
  GCNX_REQUIRE(n >= 0 && keep > 0.0f && (n == 0 || (y && dy && dx)), GCNX_ERR_INVALID_ARG,
               "relu_dropout_bwd_philox: bad argument");
  if (n == 0) return GCNX_OK;
  const int vec_ok = aligned16(y) && aligned16(dy) && aligned16(dx) && (offset % 4 == 0);
  hipLaunchKernelGGL((dropout_philox_kernel<1>), dim3(stream_grid(n / 4 + 1)), dim3(kBlock), 0,
                     as_stream(stream), n, dy, y, keep, 1.0f / keep, seed, offset, dx,
                     (uint8_t *)nullptr, vec_ok);
  GCNX_CHECK_LAUNCH();
  return GCNX_OK;
}

extern "C" int gcnx_normalize_csr_f64(const gcnx_csr *a, int64_t row_begin, const int32_t *row_sum,
                                      const double *d_table, int64_t d_table_len, float *val_out,
                                      int *status_flag, void *stream) {
  GCNX_REQUIRE(a && row_sum && d_table && val_out && status_flag && d_table_len > 0 && row_begin >= 0,
               GCNX_ERR_INVALID_ARG, "normalize: bad argument");
  if (a->nnz == 0) return GCNX_OK;
  hipLaunchKernelGGL(normalize_kernel, dim3((unsigned)ceil_div(a->nnz, kBlock)), dim3(kBlock), 0,
                     as_stream(stream), a->nnz, a->n_rows, row_begin, a->row_ptr, a->col_idx, a->val,
                     row_sum, d_table, d_table_len, val_out, status_flag);
  GCNX_CHECK_LAUNCH();
  return GCNX_OK;
}

extern "C" int gcnx_coo_interleave_supports(int32_t n_sup, const int64_t *h_nnz,
                                            const int32_t *const *h_row,
                                            const int32_t *const *h_col, int64_t coord_stride,
                                            const void *const *h_val, int val_is_f64,
                                            int32_t *out_row, int32_t *out_col, void *out_val,
                                            void *stream) {
  GCNX_REQUIRE(n_sup > 0 && h_nnz && h_row && h_col && h_val && coord_stride >= 1,
               GCNX_ERR_INVALID_ARG, "interleave: bad argument");
  int64_t off = 0;
  for (int32_t i = 0; i < n_sup; ++i) {
    const int64_t nnz = h_nnz[i];
    GCNX_REQUIRE(nnz >= 0, GCNX_ERR_INVALID_ARG, "interleave: negative nnz");
    if (nnz == 0) continue;
    GCNX_REQUIRE(h_row[i] && h_col[i] && h_val[i] && out_row && out_col && out_val,
                 GCNX_ERR_INVALID_ARG, "interleave: null array");
    void *ov = val_is_f64 ? static_cast<void *>(static_cast<double *>(out_val) + off)
                          : static_cast<void *>(static_cast<float *>(out_val) + off);
    hipLaunchKernelGGL(interleave_kernel, dim3((unsigned)ceil_div(nnz, kBlock)), dim3(kBlock), 0,
                       as_stream(stream), nnz, n_sup, i, h_row[i], h_col[i], coord_stride, h_val[i],
                       val_is_f64, out_row + off, out_col + off, ov);
    GCNX_CHECK_LAUNCH();
    off += nnz;
  }
  return GCNX_OK;
}
