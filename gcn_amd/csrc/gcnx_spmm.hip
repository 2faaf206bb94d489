// CSR SpMM for gfx950:  C = act(alpha * A' * B + beta * C)
//
// Work decomposition (merge-path flavoured row split): the nonzeros are cut into
// equal chunks (64..512, by matrix size); ONE wavefront owns one chunk, whatever rows it covers,
// so a hub row with 10^5 nonzeros costs the same per wavefront as 10^5 short
// rows.  Inside a chunk the wavefront walks the nonzeros in slot order; lane l
// owns VEC consecutive output columns, so one nonzero is one coalesced
// 64*VEC*4-byte read of a B row and VEC FMAs per lane, and a row's sum is
// formed sequentially in registers (no cross-lane reduction at this width).
// col_idx / val are read 64 at a time, one per lane (coalesced), and broadcast
// with v_readlane, which makes the B row address scalar: the gather is
// `global_load_dwordx4 v, v_lane_off, s[row_base]`.  Row ends come from a
// lane-held window of 64 row_ptr entries, so all control flow is scalar.
//
// Rows that lie wholly inside a chunk are stored straight to C.  A row cut by a
// chunk boundary leaves one partial sum per chunk it touches in `partial`
// (slot 0: the row began in an earlier chunk; slot 1: it begins here and runs
// on); spmm_fixup_kernel adds them in chunk order and applies the epilogue, so
// the result does not depend on scheduling.
#include <atomic>
#include <mutex>
#include <type_traits>

#include "gcnx_common.h"
#include "gcnx_philox.h"

namespace gcnx {

template <int VEC> struct vec_of;
template <> struct vec_of<1> { typedef float type; };
template <> struct vec_of<2> { typedef float __attribute__((ext_vector_type(2))) type; };
template <> struct vec_of<4> { typedef float __attribute__((ext_vector_type(4))) type; };

template <int VEC>
__device__ __forceinline__ typename vec_of<VEC>::type vload(const float *p) {
  return *reinterpret_cast<const typename vec_of<VEC>::type *>(p);
}
template <int VEC>
__device__ __forceinline__ void vstore(float *p, typename vec_of<VEC>::type v) {
  *reinterpret_cast<typename vec_of<VEC>::type *>(p) = v;
}
template <int VEC>
__device__ __forceinline__ typename vec_of<VEC>::type vmax0(typename vec_of<VEC>::type v) {
  // compare-and-select, not fmaxf: a NaN sum stays a NaN (as relu does in the
  // reference's framework), so that a diverged run shows up in the loss
  if constexpr (VEC == 1) {
    return v < 0.0f ? 0.0f : v;
  } else {
    typename vec_of<VEC>::type o;
#pragma unroll
    for (int i = 0; i < VEC; ++i) o[i] = v[i] < 0.0f ? 0.0f : v[i];
    return o;
  }
}

__device__ __forceinline__ int rl_i(int v, int lane) {
  return __builtin_amdgcn_readlane(v, lane);
}
__device__ __forceinline__ float rl_f(float v, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}

constexpr int kRowEndBeyond = 0x7fffffff;

struct SpmmArgs {
  int64_t n_rows;
  int64_t nnz;
  int64_t n_chunks;   // chunks of `chunk` nonzeros
  int32_t chunk;      // nonzeros one wavefront owns: kChunkUnit * chunk_mul
  int32_t chunk_mul;  // chunk_row is tabulated every kChunkUnit nonzeros
  const int64_t *row_ptr;
  const int32_t *col_idx;
  const float *val;
  const int32_t *chunk_row;
  // nullable cache hint of the wide kernel: bit (k % 64) of word k / 64 clear <=> the
  // operand row of nonzero k is read with a streaming (non-temporal) load
  const unsigned long long *keep_bits;
  const float *B;
  int64_t ldb;
  const float *B2;   // rows >= split come from here (halo rows received from peers)
  int64_t ldb2;
  int32_t split;     // INT32_MAX when there is no second operand
  float *C;
  int64_t ldc;
  int32_t width;
  int32_t pwidth;  // leading dimension of a partial slot
  float alpha;
  float beta;
  const uint8_t *vmask;
  const int64_t *mindex;
  float mscale;
  int relu;
  float *partial;
  // dropout fused behind the activation (keep >= 1: off).  Element (r, col) of
  // the output has global index drop_offset + r * drop_ld + drop_col0 + col.
  float drop_keep;
  uint64_t drop_seed;
  uint64_t drop_offset;
  int64_t drop_ld;
  int64_t drop_col0;
  // nullable: output row r draws the mask of row drop_rows[r] (the output holds a
  // subset of the rows of the activation the mask is defined on)
  const int32_t *drop_rows;
  // nullable (wide kernel only): sign pattern of the stored output, bit (c % 32) of
  // word sign_bits[r * ld_sign + c / 32] set <=> stored element (r, c) > 0
  uint32_t *sign_bits;
  int64_t ld_sign;
  // factored dense operand (spmm_factored_kernel): row c of B is not stored but
  // formed as  gate(c, j) ? fscale * sum_k D[c, k] * W(k, j) : 0  with
  // W(k, j) = Wm[k * w_sk + j * w_sj]; rows >= split come from D2 / gbits2
  const float *D;
  int64_t ldd;
  const float *D2;
  int64_t ldd2;
  const float *Wm;
  int64_t w_sk, w_sj;
  const uint32_t *gbits;
  int64_t ldgb;
  const uint32_t *gbits2;
  int64_t ldgb2;
  float fscale;
};

template <int VEC>
__device__ __forceinline__ typename vec_of<VEC>::type epilogue_store(const SpmmArgs &a, int64_t r, int col0,
                                                                     typename vec_of<VEC>::type acc) {
  float *dst = a.C + r * a.ldc + col0;
  typename vec_of<VEC>::type out = acc * a.alpha;
  if (a.beta != 0.0f) out += vload<VEC>(dst) * a.beta;
  if (a.relu) out = vmax0<VEC>(out);
  if (a.drop_keep < 1.0f) {
    const uint64_t dr = a.drop_rows ? (uint64_t)a.drop_rows[r] : (uint64_t)r;
    const uint64_t e0 = a.drop_offset + dr * (uint64_t)a.drop_ld + (uint64_t)(a.drop_col0 + col0);
    const float inv = 1.0f / a.drop_keep;
    if constexpr (VEC == 4) {
      // the host checked that e0 is a multiple of 4: one Philox block per store
      const philox4 rnd = philox4x32_10(e0 >> 2, 0ull, a.drop_seed);
#pragma unroll
      for (int i = 0; i < 4; ++i)
        out[i] = ((float)(rnd.v[i] >> 8) * (1.0f / 16777216.0f) < a.drop_keep) ? out[i] * inv : 0.0f;
    } else {
      const philox4 rnd = philox4x32_10(e0 >> 2, 0ull, a.drop_seed);
      out = ((float)(rnd.v[e0 & 3] >> 8) * (1.0f / 16777216.0f) < a.drop_keep) ? out * inv : 0.0f;
    }
  }
  vstore<VEC>(dst, out);
  return out;
}

// Lane l holds the stored columns 4l..4l+3 of row r; the 8 lanes of one 32-column
// word OR their nibbles together with three DPP moves (no LDS traffic) and the
// first of them stores the word.  Called with whole groups of 8 lanes active
// (width % 32 == 0).
__device__ __forceinline__ void store_sign_word(const SpmmArgs &a, int64_t r, int col0, int lane,
                                                typename vec_of<4>::type v) {
  int w = ((v[0] > 0.0f ? 1 : 0) | (v[1] > 0.0f ? 2 : 0) | (v[2] > 0.0f ? 4 : 0) | (v[3] > 0.0f ? 8 : 0))
          << (4 * (lane & 7));
  w |= __builtin_amdgcn_mov_dpp(w, 0xB1, 0xF, 0xF, true);    // quad_perm [1,0,3,2]
  w |= __builtin_amdgcn_mov_dpp(w, 0x4E, 0xF, 0xF, true);    // quad_perm [2,3,0,1]
  w |= __builtin_amdgcn_mov_dpp(w, 0x141, 0xF, 0xF, true);   // row_half_mirror: the other quad of the 8
  if ((lane & 7) == 0) a.sign_bits[r * a.ld_sign + (col0 >> 5)] = (uint32_t)w;
}

#ifndef GCNX_WIDE_U
#define GCNX_WIDE_U 8       // gathers in flight per wavefront in the wide kernel
#endif
#ifndef GCNX_WIDE_WAVES
#define GCNX_WIDE_WAVES 4   // wavefronts per SIMD the wide kernel is register-budgeted for
#endif

template <int VEC, int U>
__global__ __launch_bounds__(kBlock, GCNX_WIDE_WAVES) void spmm_chunk_kernel(const SpmmArgs a) {
  typedef typename vec_of<VEC>::type V;
  const int lane = threadIdx.x & (kWave - 1);
  const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int64_t c = (int64_t)blockIdx.x * kWavesPerBlock + wid;
  if (c >= a.n_chunks) return;
  const int col0 = (int)blockIdx.y * (kWave * VEC) + lane * VEC;
  const bool active = col0 < a.width;

  const int chunk = a.chunk;
  const int64_t n0 = c * (int64_t)chunk;
  const int64_t left = a.nnz - n0;
  const int nk = left < (int64_t)chunk ? (int)left : chunk;

  int64_t r = a.chunk_row[c * a.chunk_mul];
  int64_t rbase = r;
  // Lane-held window: end (relative to n0) of rows rbase .. rbase+63.
  auto load_row_ends = [&](int64_t rb) -> int {
    const int64_t rr = rb + 1 + lane;
    if (rr > a.n_rows) return kRowEndBeyond;
    const int64_t e = a.row_ptr[rr] - n0;
    return e > (int64_t)chunk ? chunk + 1 : (int)e;
  };
  int re_v = load_row_ends(rbase);
  int re = rl_i(re_v, 0);
  int rs;  // start of row r relative to n0 (negative: began in an earlier chunk)
  {
    const int64_t s = a.row_ptr[r] - n0;
    rs = s < 0 ? -1 : (int)s;
  }

  V acc = V(0.0f);
  float *const pslot = a.partial + (size_t)c * 2 * a.pwidth + col0;

  auto close_row = [&]() {
    if (active) {
      if (rs < 0) {
        vstore<VEC>(pslot, acc);
      } else {
        const V out = epilogue_store<VEC>(a, r, col0, acc);
        if constexpr (VEC == 4) {
          if (a.sign_bits) store_sign_word(a, r, col0, lane, out);
        }
      }
    }
    acc = V(0.0f);
    rs = re;
    ++r;
    int idx = (int)(r - rbase);
    if (idx == kWave) {
      rbase = r;
      re_v = load_row_ends(rbase);
      idx = 0;
    }
    re = rl_i(re_v, idx);
  };

  auto load_batch = [&](int kb, int &cols, float &vals) {
    const int k = kb + lane;
    cols = 0;
    vals = 0.0f;
    if (k < nk) {
      const int64_t g = n0 + k;
      cols = a.col_idx[g];
      float v = a.val[g];
      if (a.vmask) {
        const int64_t mi = a.mindex ? a.mindex[g] : g;
        v = a.vmask[mi] ? v * a.mscale : 0.0f;
      }
      vals = v;
    }
  };

  int cols_n;
  float vals_n;
  load_batch(0, cols_n, vals_n);

  for (int kb = 0; kb < nk; kb += kWave) {
    const int cols = cols_n;
    const float vals = vals_n;
    if (kb + kWave < nk) load_batch(kb + kWave, cols_n, vals_n);
    const int jn = (nk - kb) < kWave ? (nk - kb) : kWave;
    // keep bits of this batch of 64 nonzeros (chunks start at multiples of 64): one
    // wavefront-uniform word; without a hint every row is kept
    unsigned long long keep = ~0ull;
    if (a.keep_bits) {
      const unsigned long long w = a.keep_bits[(n0 + kb) >> 6];
      keep = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(w >> 32)) << 32) |
             (unsigned)__builtin_amdgcn_readfirstlane((int)w);
    }
    for (int j = 0; j < jn; j += U) {
      V b[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int jj = (j + u) < jn ? (j + u) : (jn - 1);
        const int cc = rl_i(cols, jj);
        const float *src = (cc < a.split ? a.B + (int64_t)cc * a.ldb
                                         : a.B2 + (int64_t)(cc - a.split) * a.ldb2) + col0;
        b[u] = V(0.0f);
        if (active) {
          if ((keep >> jj) & 1ull) {
            b[u] = vload<VEC>(src);
          } else {
            // a rarely referenced row: streamed, so that it does not evict the rows worth keeping
            const float *cold = src;
            asm volatile("" : "+v"(cold));      // keeps the two loads (and their cache policies) apart
            b[u] = __builtin_nontemporal_load(reinterpret_cast<const V *>(cold));
          }
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (j + u < jn) {
          const int k = kb + j + u;
          while (k == re) close_row();
          const float v = rl_f(vals, j + u);
          acc += b[u] * v;
        }
      }
    }
  }
  while (re <= nk) close_row();
  if (r < a.n_rows && rs < nk && active) {
    vstore<VEC>(pslot + (rs < 0 ? 0 : a.pwidth), acc);
  }
}

// The wide kernel with a FACTORED dense operand: B = gate .* (D * W) * fscale with
// D [n_cols, 16] and W [16, width] (the backward pass of a layer that follows a
// relu + dropout: B = d(loss)/d(H') gated by the sign pattern of H').  B is never
// materialised: for every nonzero the wavefront reads the 64-byte row of D with
// one scalar load (the column index is wavefront-uniform) and the 32-bit gate
// word of its columns, and forms its four elements of the row with 64 FMAs
// against weights held in registers -- 96 bytes fetched per nonzero instead of a
// 1-KiB row, paid for with vector arithmetic the memory-bound kernel has to
// spare.  The sum over k is a chain of fused multiply-adds from zero in the
// order in which the two-kernel path feeds k to its matrix instruction; the gate
// scale multiplies the matrix value instead of the element, so both paths give
// the same bits when the scale is a power of two (dropout 0.5) and differ by one
// rounding per term otherwise.  This is the fallback of spmm_factored_mfma_kernel
// for operands whose pitch rules out 16-byte loads.
constexpr int kFactoredInner = 16;
typedef float __attribute__((ext_vector_type(16))) f16s;
typedef const f16s __attribute__((address_space(4))) *const_f16s_ptr;

template <int U>
__global__ __launch_bounds__(kBlock, 2) void spmm_factored_kernel(const SpmmArgs a) {
  typedef typename vec_of<4>::type V;
  const int lane = threadIdx.x & (kWave - 1);
  const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int64_t c = (int64_t)blockIdx.x * kWavesPerBlock + wid;
  if (c >= a.n_chunks) return;
  const int col0 = (int)blockIdx.y * (kWave * 4) + lane * 4;
  const bool active = col0 < a.width;

  const int chunk = a.chunk;
  const int64_t n0 = c * (int64_t)chunk;
  const int64_t left = a.nnz - n0;
  const int nk = left < (int64_t)chunk ? (int)left : chunk;

  // weights of this lane's four columns
  V w[kFactoredInner];
#pragma unroll
  for (int k = 0; k < kFactoredInner; ++k) {
#pragma unroll
    for (int j = 0; j < 4; ++j) w[k][j] = active ? a.Wm[k * a.w_sk + (col0 + j) * a.w_sj] : 0.0f;
  }
  const int word = col0 >> 5, shift = col0 & 31;

  int64_t r = a.chunk_row[c * a.chunk_mul];
  int64_t rbase = r;
  auto load_row_ends = [&](int64_t rb) -> int {
    const int64_t rr = rb + 1 + lane;
    if (rr > a.n_rows) return kRowEndBeyond;
    const int64_t e = a.row_ptr[rr] - n0;
    return e > (int64_t)chunk ? chunk + 1 : (int)e;
  };
  int re_v = load_row_ends(rbase);
  int re = rl_i(re_v, 0);
  int rs;
  {
    const int64_t s0 = a.row_ptr[r] - n0;
    rs = s0 < 0 ? -1 : (int)s0;
  }
  V acc = V(0.0f);
  float *const pslot = a.partial + (size_t)c * 2 * a.pwidth + col0;

  auto close_row = [&]() {
    if (active) {
      if (rs < 0) vstore<4>(pslot, acc);
      else vstore<4>(a.C + r * a.ldc + col0, acc);
    }
    acc = V(0.0f);
    rs = re;
    ++r;
    int idx = (int)(r - rbase);
    if (idx == kWave) {
      rbase = r;
      re_v = load_row_ends(rbase);
      idx = 0;
    }
    re = rl_i(re_v, idx);
  };

  auto load_batch = [&](int kb, int &cols, float &vals) {
    const int k = kb + lane;
    cols = 0;
    vals = 0.0f;
    if (k < nk) {
      cols = a.col_idx[n0 + k];
      vals = a.val[n0 + k];
    }
  };
  int cols_n;
  float vals_n;
  load_batch(0, cols_n, vals_n);

  for (int kb = 0; kb < nk; kb += kWave) {
    const int cols = cols_n;
    const float vals = vals_n;
    if (kb + kWave < nk) load_batch(kb + kWave, cols_n, vals_n);
    const int jn = (nk - kb) < kWave ? (nk - kb) : kWave;
    for (int j = 0; j < jn; j += U) {
      f16s d[U];
      uint32_t gw[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int jj = (j + u) < jn ? (j + u) : (jn - 1);
        const int cc = rl_i(cols, jj);
        const bool far = cc >= a.split;
        const float *drow = far ? a.D2 + (int64_t)(cc - a.split) * a.ldd2 : a.D + (int64_t)cc * a.ldd;
        const uint32_t *grow = far ? a.gbits2 + (int64_t)(cc - a.split) * a.ldgb2 : a.gbits + (int64_t)cc * a.ldgb;
        // constant address space + uniform address: one s_load_dwordx16
        d[u] = *reinterpret_cast<const_f16s_ptr>(reinterpret_cast<uintptr_t>(drow));
        gw[u] = active ? grow[word] : 0u;
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (j + u < jn) {
          const int k = kb + j + u;
          while (k == re) close_row();
          const float v = rl_f(vals, j + u) * a.fscale;   // the gate scale rides on the matrix value
          V t = V(0.0f);
#pragma unroll
          for (int q = 0; q < kFactoredInner; ++q) {
            // the order in which gemm_small_k_mfma_kernel feeds k to the matrix
            // instruction (step s takes k = 8 (s / 4) + s % 4 and that + 4)
            const int kk = 8 * (q >> 3) + 4 * (q & 1) + ((q >> 1) & 3);
#pragma unroll
            for (int jx = 0; jx < 4; ++jx) t[jx] = __builtin_fmaf(d[u][kk], w[kk][jx], t[jx]);
          }
          const uint32_t bits = gw[u] >> shift;
          V g;
#pragma unroll
          for (int jx = 0; jx < 4; ++jx) g[jx] = ((bits >> jx) & 1u) ? t[jx] : 0.0f;
          acc += g * v;
        }
      }
    }
  }
  while (re <= nk) close_row();
  if (r < a.n_rows && rs < nk && active) vstore<4>(pslot + (rs < 0 ? 0 : a.pwidth), acc);
}

// The factored product with the row-forming on the matrix pipe.  The two halves
// of a wavefront each own one chunk.  Per round each half contributes 16 of its
// nonzeros as rows of a 32 x 16 A operand (their D rows); eight chains of
// v_mfma_f32_32x32x2_f32 against the weights give T = D_rows * W for 32 nonzeros
// x 256 columns.  In the C/D layout a lane holds one column per tile and, in its
// 16 registers, the 16 nonzeros of ITS half in order (MFMA row i belongs to half
// (i >> 2) & 1 and is that half's nonzero (i & 3) + 4 (i >> 3)), so each half
// then walks its nonzeros exactly as the wide kernel does -- gate, scale by the
// matrix value, add to the open row, close rows at their ends -- with per-half
// (divergent) row bookkeeping.  The k order of the chains is the one of
// gemm_small_k_mfma_kernel and rows are summed chunk-sequentially; the gate scale
// multiplies the matrix value instead of the element, so the result has the bits
// of the two-kernel path when the scale is a power of two (dropout 0.5) and
// differs from it by one rounding per term otherwise.
typedef float __attribute__((ext_vector_type(16))) f16acc;
typedef uint32_t __attribute__((ext_vector_type(4))) u4v;

// NT: 32-column tiles per wavefront (a slab of 32 NT columns per blockIdx.y);
// FULL: the width is a multiple of the slab, no tile needs a bounds check.
//
// SPLIT: the row-forming runs on the bf16 matrix pipe at fp32 accuracy instead of
// the f32-input MFMA (which runs at the vector rate and does not overlap with the
// vector instructions of the update): D rows and weights are split exactly into
// three bf16 pieces each, x = h + m + l, and T is the sum of the six partial
// products of weight >= 2^-16 (what is dropped is below 2^-23 of each product, one
// fp32 rounding), six v_mfma_f32_32x32x16_bf16 per tile instead of eight chained
// v_mfma_f32_32x32x2_f32.  The top pieces of the weights stay in registers, the
// two lower ones are shared by the workgroup's wavefronts through LDS.  Results
// then agree with the two-kernel path to fp32 rounding, not to the bit.
typedef __bf16 __attribute__((ext_vector_type(8))) bf8s;
typedef int __attribute__((ext_vector_type(4))) i4s;

__device__ __forceinline__ void split3f(float x, unsigned &h, unsigned &m, unsigned &l) {
  h = __float_as_uint(x) & 0xffff0000u;
  const float r = x - __uint_as_float(h);
  m = __float_as_uint(r) & 0xffff0000u;
  l = __float_as_uint(r - __uint_as_float(m));   // at most 8 significant bits: its lower half is zero
}
__device__ __forceinline__ int pack_up(unsigned lo, unsigned hi) {
  return (int)__builtin_amdgcn_perm(hi, lo, 0x07060302u);
}
// eight fp32 values (consecutive k) -> the three bf16 operands of one lane
__device__ __forceinline__ void split8(const float (&x)[8], i4s &oh, i4s &om, i4s &ol) {
  unsigned h[8], m[8], l[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) split3f(x[j], h[j], m[j], l[j]);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    oh[j] = pack_up(h[2 * j], h[2 * j + 1]);
    om[j] = pack_up(m[2 * j], m[2 * j + 1]);
    ol[j] = pack_up(l[2 * j], l[2 * j + 1]);
  }
}

template <int NT, bool FULL, bool SPLIT>
__global__ __launch_bounds__(kBlock, 2) void spmm_factored_mfma_kernel(const SpmmArgs a) {
  // the ragged form (a last slab of fewer tiles) exists for 4-tile slabs only: with 8
  // tiles its per-tile bounds and the per-half row state do not fit the scalar
  // registers together
  static_assert(FULL || NT <= 4, "ragged slabs are served by the 4-tile form");
  constexpr int HW = 32;     // lanes per half
  const int lane = threadIdx.x & (kWave - 1);
  const int li = lane & 31, h = lane >> 5;
  const int wid = (int)(threadIdx.x >> 6);
  const int64_t ch = 2 * ((int64_t)blockIdx.x * kWavesPerBlock + wid) + h;
  const bool live = ch < a.n_chunks;
  const int cbase = (int)blockIdx.y * (NT * 32);
  const int nt = FULL ? NT : ((a.width - cbase) / 32 < NT ? (a.width - cbase) / 32 : NT);

  // op(W) fragments: B[k][j] of tile t.  Exact form: k fed in the order of
  // gemm_small_k_mfma_kernel.  Split form: lane (li, h) holds k = 8 h .. 8 h + 7 of
  // column li of its tile, as three bf16 operands; the top one stays in registers,
  // the two lower ones go to LDS (one copy for the four wavefronts).
  float bf[SPLIT ? 1 : NT][8];
  i4s bh[SPLIT ? NT : 1];
  __shared__ __attribute__((aligned(16))) i4s wlow[SPLIT ? 2 * NT * kWave : 1];
  if constexpr (!SPLIT) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int sx = 0; sx < 8; ++sx) {
        const int k = 8 * (sx >> 2) + 4 * h + (sx & 3);
        bf[t][sx] = t < nt ? a.Wm[k * a.w_sk + (int64_t)(cbase + 32 * t + li) * a.w_sj] : 0.0f;
      }
  } else {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      float x[8];
#pragma unroll
      for (int k = 0; k < 8; ++k)
        x[k] = t < nt ? a.Wm[(8 * h + k) * a.w_sk + (int64_t)(cbase + 32 * t + li) * a.w_sj] : 0.0f;
      i4s om, ol;
      split8(x, bh[t], om, ol);
      if (threadIdx.x < kWave) {      // every wavefront computes the same fragments: one writes
        wlow[t * kWave + lane] = om;
        wlow[(NT + t) * kWave + lane] = ol;
      }
    }
    __syncthreads();
  }

  const int chunk = a.chunk;
  const int64_t n0 = live ? ch * (int64_t)chunk : 0;
  int nk = 0;
  if (live) {
    const int64_t left = a.nnz - n0;
    nk = left < (int64_t)chunk ? (int)left : chunk;
  }
  const int nk_other = __shfl_xor(nk, HW, kWave);
  // the same in every lane; held as a scalar so that the round loops are scalar loops
  const int nk_max = __builtin_amdgcn_readfirstlane(nk > nk_other ? nk : nk_other);

  // Row bookkeeping.  The two halves walk different chunks, but within a half every
  // lane is at the same row: the state of each half (open row, its start and end,
  // the output pointer) is held in scalar registers, one set per half, and a row
  // end is a scalar compare.  Only the stores and the clearing of the accumulators
  // run under the half's lane mask.
  const int r_v = live ? a.chunk_row[ch * a.chunk_mul] : (int)a.n_rows;
  // half-held windows: lane li holds the end (relative to n0) of row rb + li; the
  // window after the current one is fetched a window ahead
  // (the window ahead is kept as loaded and only turned into row ends when it
  // becomes current, so that nothing waits for the load where it is issued)
  auto load_row_ptrs = [&](int64_t rb) -> int64_t {
    const int64_t rr = rb + 1 + li;
    return rr > a.n_rows ? INT64_MAX : a.row_ptr[rr];
  };
  auto row_ends = [&](int64_t ptrs) -> int {       // beyond the chunk (or the matrix): chunk + 1
    const int64_t e = ptrs - n0;
    return e > (int64_t)chunk ? chunk + 1 : (int)e;
  };
  int re_v = row_ends(load_row_ptrs(r_v));
  int64_t nx_ptrs = load_row_ptrs((int64_t)r_v + HW);
  int rs_v = -1;
  if (live && r_v < a.n_rows) {
    const int64_t s0 = a.row_ptr[r_v] - n0;
    rs_v = s0 < 0 ? -1 : (int)s0;
  }
  const int64_t wave_s = (int64_t)blockIdx.x * kWavesPerBlock + __builtin_amdgcn_readfirstlane(wid);
  int r_s[2], rbase_s[2], re_s[2], rs_s[2];
  float *crow_s[2];
#pragma unroll
  for (int hh = 0; hh < 2; ++hh) {
    r_s[hh] = rbase_s[hh] = __builtin_amdgcn_readlane(r_v, HW * hh);
    re_s[hh] = __builtin_amdgcn_readlane(re_v, HW * hh);
    rs_s[hh] = __builtin_amdgcn_readlane(rs_v, HW * hh);
    crow_s[hh] = a.C + (int64_t)r_s[hh] * a.ldc + cbase;             // lane li: + li
  }
  // the half's partial-sum slots (first row continued from the chunk before, last
  // row continued in the chunk after); needed twice per chunk, formed on demand
  auto pslot_of = [&](int hh) -> float * {
    const int64_t ch_s = 2 * wave_s + hh;
    return a.partial + (size_t)(ch_s < a.n_chunks ? ch_s : 0) * 2 * a.pwidth + cbase;
  };
  float acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = 0.0f;

  auto close_half = [&](auto HH) {
    constexpr int hh = decltype(HH)::value;
    float *const dst = rs_s[hh] < 0 ? pslot_of(hh) : crow_s[hh];
    if (h == hh) {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        if (t < nt) dst[li + 32 * t] = acc[t];
        acc[t] = 0.0f;
      }
    }
    rs_s[hh] = re_s[hh];
    ++r_s[hh];
    crow_s[hh] += a.ldc;
    int idx = r_s[hh] - rbase_s[hh];
    if (idx == HW) {
      rbase_s[hh] = r_s[hh];
      if (h == hh) {
        re_v = row_ends(nx_ptrs);
        nx_ptrs = load_row_ptrs((int64_t)rbase_s[hh] + HW);
      }
      idx = 0;
    }
    re_s[hh] = __builtin_amdgcn_readlane(re_v, HW * hh + idx);
  };
  const std::integral_constant<int, 0> kHalf0;
  const std::integral_constant<int, 1> kHalf1;

  // MFMA row i = li of the A operand is nonzero an of half ah
  const int ah = (li >> 2) & 1, an = (li & 3) + 4 * (li >> 3);

  // T = D_rows * W for the 32 nonzeros of a round, from the A fragments (exact
  // form: af, split form: a_h / a_m / a_l)
  auto form_rows = [&](f16acc (&T)[NT], const float (&af)[8], const i4s &a_h, const i4s &a_m, const i4s &a_l) {
    // the wavefront that is on the matrix pipe goes first: resident wavefronts
    // would otherwise share the pipe evenly, stay in step and never overlap
    // one's matrix phase with the other's vector phase
    __builtin_amdgcn_s_setprio(3);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
      for (int i = 0; i < 16; ++i) T[t][i] = 0.0f;
      if (t < nt) {
        if constexpr (!SPLIT) {
#pragma unroll
          for (int sx = 0; sx < 8; ++sx) T[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[sx], bf[t][sx], T[t], 0, 0, 0);
        } else {
          // smallest partial products first
          const bf8s wm = __builtin_bit_cast(bf8s, wlow[t * kWave + lane]);
          const bf8s wl = __builtin_bit_cast(bf8s, wlow[(NT + t) * kWave + lane]);
          const bf8s wh = __builtin_bit_cast(bf8s, bh[t]);
          const bf8s xh = __builtin_bit_cast(bf8s, a_h), xm = __builtin_bit_cast(bf8s, a_m),
                     xl = __builtin_bit_cast(bf8s, a_l);
          T[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, wl, T[t], 0, 0, 0);
          T[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xm, wm, T[t], 0, 0, 0);
          T[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xl, wh, T[t], 0, 0, 0);
          T[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, wm, T[t], 0, 0, 0);
          T[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xm, wh, T[t], 0, 0, 0);
          T[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh, wh, T[t], 0, 0, 0);
        }
      }
    }
    __builtin_amdgcn_s_setprio(0);
  };
  // each half walks the 16 nonzeros k0 .. k0 + 15 of its chunk: gate, scale by the
  // matrix value, add to the open row, close rows at their ends
  auto walk = [&](const f16acc (&T)[NT], const u4v &wq, float vals, int k0, int sb) {
#pragma unroll
      for (int n = 0; n < 16; ++n) {
        const int k = k0 + n;
        // row ends of either half (scalar compares); the arithmetic below runs on
        // both halves at once.  (Past the end of a chunk k can only meet a row end
        // that lies exactly on the chunk end, which the loop after the rounds would
        // close anyway.)
        while (k == re_s[0]) close_half(kHalf0);
        while (k == re_s[1]) close_half(kHalf1);
        const int src = 16 * sb + n;
        // exact form: the half's own matrix value (LDS crossbar, no vector-ALU slot);
        // split form: the value already multiplies the D row, the update is an addition
        float vs = 1.0f;
        if constexpr (!SPLIT) vs = __shfl(vals, h * HW + src, kWave);
        // The two halves' gate words of a tile, side by side, ARE the lane mask of
        // that tile's update (bit li of the low word gates lane li, of the high
        // word lane 32 + li): the multiply-add runs under that mask as EXEC.
        uint64_t mask[NT];
        // the mask the surrounding code runs under (all lanes: the loop bounds here
        // are wavefront-uniform), read from the hardware and put back after the
        // gated updates instead of assuming all ones
        const uint64_t exec_all = __builtin_amdgcn_read_exec();
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const int wl = n + 16 * (t >> 2);
          const uint32_t m_lo = (uint32_t)rl_i((int)wq[t & 3], wl), m_hi = (uint32_t)rl_i((int)wq[t & 3], HW + wl);
          mask[t] = ((uint64_t)m_hi << 32) | m_lo;
        }
        if constexpr (SPLIT && NT == 8) {
          asm volatile(
              "s_mov_b64 exec, %16\n\tv_add_f32 %0, %0, %8\n\t"
              "s_mov_b64 exec, %17\n\tv_add_f32 %1, %1, %9\n\t"
              "s_mov_b64 exec, %18\n\tv_add_f32 %2, %2, %10\n\t"
              "s_mov_b64 exec, %19\n\tv_add_f32 %3, %3, %11\n\t"
              "s_mov_b64 exec, %20\n\tv_add_f32 %4, %4, %12\n\t"
              "s_mov_b64 exec, %21\n\tv_add_f32 %5, %5, %13\n\t"
              "s_mov_b64 exec, %22\n\tv_add_f32 %6, %6, %14\n\t"
              "s_mov_b64 exec, %23\n\tv_add_f32 %7, %7, %15\n\t"
              "s_mov_b64 exec, %24"
              : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]),
                "+v"(acc[7])
              : "v"(T[0][n]), "v"(T[1][n]), "v"(T[2][n]), "v"(T[3][n]), "v"(T[4][n]), "v"(T[5][n]), "v"(T[6][n]),
                "v"(T[7][n]), "s"(mask[0]), "s"(mask[1]), "s"(mask[2]), "s"(mask[3]), "s"(mask[4]), "s"(mask[5]),
                "s"(mask[6]), "s"(mask[7]), "s"(exec_all));
        } else if constexpr (SPLIT) {
          asm volatile(
              "s_mov_b64 exec, %8\n\tv_add_f32 %0, %0, %4\n\t"
              "s_mov_b64 exec, %9\n\tv_add_f32 %1, %1, %5\n\t"
              "s_mov_b64 exec, %10\n\tv_add_f32 %2, %2, %6\n\t"
              "s_mov_b64 exec, %11\n\tv_add_f32 %3, %3, %7\n\t"
              "s_mov_b64 exec, %12"
              : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3])
              : "v"(T[0][n]), "v"(T[1][n]), "v"(T[2][n]), "v"(T[3][n]), "s"(mask[0]), "s"(mask[1]), "s"(mask[2]),
                "s"(mask[3]), "s"(exec_all));
        } else if constexpr (NT == 8) {
          asm volatile(
              "s_mov_b64 exec, %16\n\tv_fmac_f32 %0, %8, %24\n\t"
              "s_mov_b64 exec, %17\n\tv_fmac_f32 %1, %9, %24\n\t"
              "s_mov_b64 exec, %18\n\tv_fmac_f32 %2, %10, %24\n\t"
              "s_mov_b64 exec, %19\n\tv_fmac_f32 %3, %11, %24\n\t"
              "s_mov_b64 exec, %20\n\tv_fmac_f32 %4, %12, %24\n\t"
              "s_mov_b64 exec, %21\n\tv_fmac_f32 %5, %13, %24\n\t"
              "s_mov_b64 exec, %22\n\tv_fmac_f32 %6, %14, %24\n\t"
              "s_mov_b64 exec, %23\n\tv_fmac_f32 %7, %15, %24\n\t"
              "s_mov_b64 exec, %25"
              : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]), "+v"(acc[4]), "+v"(acc[5]), "+v"(acc[6]),
                "+v"(acc[7])
              : "v"(T[0][n]), "v"(T[1][n]), "v"(T[2][n]), "v"(T[3][n]), "v"(T[4][n]), "v"(T[5][n]), "v"(T[6][n]),
                "v"(T[7][n]), "s"(mask[0]), "s"(mask[1]), "s"(mask[2]), "s"(mask[3]), "s"(mask[4]), "s"(mask[5]),
                "s"(mask[6]), "s"(mask[7]), "v"(vs), "s"(exec_all));
        } else {
          asm volatile(
              "s_mov_b64 exec, %8\n\tv_fmac_f32 %0, %4, %12\n\t"
              "s_mov_b64 exec, %9\n\tv_fmac_f32 %1, %5, %12\n\t"
              "s_mov_b64 exec, %10\n\tv_fmac_f32 %2, %6, %12\n\t"
              "s_mov_b64 exec, %11\n\tv_fmac_f32 %3, %7, %12\n\t"
              "s_mov_b64 exec, %13"
              : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3])
              : "v"(T[0][n]), "v"(T[1][n]), "v"(T[2][n]), "v"(T[3][n]), "s"(mask[0]), "s"(mask[1]), "s"(mask[2]),
                "s"(mask[3]), "v"(vs), "s"(exec_all));
        }
      }
  };

  if constexpr (SPLIT) {
    // The gathers of round j + 1 (per nonzero the 64-byte row of D and the slab's
    // gate words) are issued at the top of round j and land, by LDS-DMA
    // (global_load_lds_dwordx4: no registers), in one of two wavefront-private
    // landing slots; a lane reads back exactly the bytes it asked for, so no
    // barrier and no layout is involved -- LDS is a register-free prefetch buffer.
    // The DMA instructions are inline assembly (the compiler inserts no waits of
    // its own for them); the wait for round j's data is written out: the three
    // instructions of round j + 1 are the newest in the wavefront's in-order
    // queue, so "at most 3 outstanding" means round j has landed.  col_idx / val
    // travel one block of 32 nonzeros ahead in registers.
    __shared__ __attribute__((aligned(16))) i4s land[kWavesPerBlock][2][3 * kWave];
    i4s *const my_land = &land[wid][0][0];
    const unsigned land_lds = (unsigned)(size_t)((__attribute__((address_space(3))) i4s *)my_land);
    const int gq = 4 * (li >> 4) < nt ? (li >> 4) : 0;   // gate words 4 gq .. 4 gq + 3 of the slab (clamped: masked below)
    auto load_block = [&](int kb, int &cols, float &vals) {
      cols = 0;
      vals = 0.0f;
      if (kb + li < nk) {
        cols = a.col_idx[n0 + kb + li];
        vals = a.val[n0 + kb + li] * a.fscale;   // the gate scale rides on the matrix value
      }
    };
    // operand bases and pitches as opaque scalars (otherwise the per-lane choice
    // between the two operands becomes a choice between two kernel-argument
    // addresses followed by dependent loads, inside the loop)
    const float *d1 = a.D, *d2 = a.D2;
    const uint32_t *g1 = a.gbits, *g2 = a.gbits2;
    int64_t ld1 = a.ldd, ld2 = a.ldd2, lg1 = a.ldgb, lg2 = a.ldgb2;
    int split = a.split;
    asm volatile("" : "+s"(d1), "+s"(d2), "+s"(g1), "+s"(g2));
    asm volatile("" : "+s"(ld1), "+s"(ld2), "+s"(lg1), "+s"(lg2), "+s"(split));
    auto issue = [&](int cols, int sb, int slot) {
      const int cc = __shfl(cols, ah * HW + 16 * sb + an, kWave);
      const float *r1 = d1 + (int64_t)cc * ld1, *r2 = d2 + (int64_t)(cc - split) * ld2;
      const float *drow = (cc < split ? r1 : r2) + 8 * h;
      const int cw = __shfl(cols, h * HW + 16 * sb + (li & 15), kWave);
      const uint32_t *w1 = g1 + (int64_t)cw * lg1, *w2 = g2 + (int64_t)(cw - split) * lg2;
      const uint32_t *grow = (cw < split ? w1 : w2) + (cbase >> 5) + 4 * gq;
      const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)(land_lds + (unsigned)(slot * 3 * kWave * 16)));
      asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off\n\t"
                   "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\t"
                   "s_add_u32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off"
                   ::"v"(drow), "v"(drow + 4), "v"(grow), "s"(dst)
                   : "memory", "scc");   // s_add_u32 writes SCC (M0 is reserved: the compiler sets it before each use of its own)
    };
    int cols, cols_n = 0;
    float vals, vals_n = 0.0f;
    load_block(0, cols, vals);
    issue(cols, 0, 0);
    int slot = 0;
    for (int kb = 0; kb < nk_max; kb += HW) {
#pragma unroll 1
      for (int sb = 0; sb < 2; ++sb) {
        const int k0 = kb + 16 * sb;
        if (k0 >= nk_max) break;
        if (k0 + 16 < nk_max) {
          issue(sb == 0 ? cols : cols_n, sb ^ 1, slot ^ 1);
          asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        } else {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        // the next block's col_idx / val go out behind the wait (they are first
        // needed a round from now; ahead of it the wait would cover them too)
        if (sb == 0 && kb + HW < nk_max) load_block(kb + HW, cols_n, vals_n);
        const i4s *src = my_land + slot * 3 * kWave + lane;
        const typename vec_of<4>::type a0 = __builtin_bit_cast(typename vec_of<4>::type, src[0]),
                                       a1 = __builtin_bit_cast(typename vec_of<4>::type, src[kWave]);
        u4v wq = __builtin_bit_cast(u4v, src[2 * kWave]);
        // a nonzero past the end of the chunk (and a lane whose words lie outside
        // the slab) gets zero words, i.e. contributes an exact zero
        if (!(4 * (li >> 4) < nt && k0 + (li & 15) < nk)) wq = u4v(0u);
        // the matrix value (times the gate scale) of this lane's nonzero multiplies its
        // D row here, once per round, so that the walk below only adds
        const float mv = __shfl(vals, ah * HW + 16 * sb + an, kWave);
        float af[8];
#pragma unroll
        for (int jx = 0; jx < 4; ++jx) { af[jx] = a0[jx] * mv; af[4 + jx] = a1[jx] * mv; }
        i4s a_h, a_m, a_l;
        split8(af, a_h, a_m, a_l);
        f16acc T[NT];
        form_rows(T, af, a_h, a_m, a_l);
        walk(T, wq, vals, k0, sb);
        slot ^= 1;
      }
      cols = cols_n;
      vals = vals_n;
    }
  } else {
    for (int kb = 0; kb < nk_max; kb += HW) {
      int cols = 0;
      float vals = 0.0f;
      if (kb + li < nk) {
        cols = a.col_idx[n0 + kb + li];
        // the gate scale rides on the matrix value (one multiplication per nonzero
        // instead of one per element)
        vals = a.val[n0 + kb + li] * a.fscale;
      }
#pragma unroll 1
      for (int sb = 0; sb < 2; ++sb) {
        const int k0 = kb + 16 * sb;
        if (k0 >= nk_max) break;
        // A fragments: D row of nonzero (ah, an) of this round, k = 4 h .. and 8 + 4 h ..
        float af[8];
        i4s a_h = {0, 0, 0, 0}, a_m = a_h, a_l = a_h;
        {
          const int cc = __shfl(cols, ah * HW + 16 * sb + an, kWave);
          const float *drow = cc < a.split ? a.D + (int64_t)cc * a.ldd : a.D2 + (int64_t)(cc - a.split) * a.ldd2;
          const typename vec_of<4>::type a0 = vload<4>(drow + 4 * h), a1 = vload<4>(drow + 8 + 4 * h);
#pragma unroll
          for (int jx = 0; jx < 4; ++jx) { af[jx] = a0[jx]; af[4 + jx] = a1[jx]; }
        }
        // gate words of this half's 16 nonzeros: lane li holds words 4 g .. 4 g + 3
        // (g = li >> 4) of the slab's words of nonzero li & 15; a nonzero past the
        // end of the chunk gets zero words, i.e. contributes an exact zero
        u4v wq = u4v(0u);
        {
          const int cw = __shfl(cols, h * HW + 16 * sb + (li & 15), kWave);
          const int g = li >> 4;
          if (4 * g < nt && k0 + (li & 15) < nk) {
            const uint32_t *grow = cw < a.split ? a.gbits + (int64_t)cw * a.ldgb
                                                : a.gbits2 + (int64_t)(cw - a.split) * a.ldgb2;
            wq = *reinterpret_cast<const u4v *>(grow + (cbase >> 5) + 4 * g);
          }
        }
        f16acc T[NT];
        form_rows(T, af, a_h, a_m, a_l);
        walk(T, wq, vals, k0, sb);
      }
    }
  }
  auto finish_half = [&](auto HH) {
    constexpr int hh = decltype(HH)::value;
    if (2 * wave_s + hh >= a.n_chunks) return;
    const int nk_h = __builtin_amdgcn_readlane(nk, HW * hh);
    while (re_s[hh] <= nk_h) close_half(HH);
    if (r_s[hh] < a.n_rows && rs_s[hh] < nk_h) {
      float *const dst = pslot_of(hh) + (rs_s[hh] < 0 ? 0 : a.pwidth);
      if (h == hh) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
          if (t < nt) dst[li + 32 * t] = acc[t];
      }
    }
  };
  finish_half(kHalf0);
  finish_half(kHalf1);
}

// Narrow dense operands (at most 32 lanes wide): the wavefront is cut into
// S = 64 / LW slots of LW lanes; slot s takes nonzero k0 + s of each group of S
// consecutive nonzeros, so one load instruction gathers S different rows of B
// (each LW*VEC*4 bytes) instead of leaving most lanes idle.  A slot accumulates
// the terms of the current row that fall to it; when a row ends, the S slot sums
// are added with wavefront shuffles (xor over the slot bits) and slot 0 stores.
// Row ends inside a group are handled by range-masking the group's terms, all
// driven by scalar comparisons.
template <int VEC, int LW, int U>
__global__ __launch_bounds__(kBlock) void spmm_slot_kernel(const SpmmArgs a) {
  typedef typename vec_of<VEC>::type V;
  constexpr int S = kWave / LW;
  const int lane = threadIdx.x & (kWave - 1);
  const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int64_t c = (int64_t)blockIdx.x * kWavesPerBlock + wid;
  if (c >= a.n_chunks) return;
  const int slot = lane / LW;
  const int col0 = (lane % LW) * VEC;
  const bool active = col0 < a.width;

  const int chunk = a.chunk;
  const int64_t n0 = c * (int64_t)chunk;
  const int64_t left = a.nnz - n0;
  const int nk = left < (int64_t)chunk ? (int)left : chunk;

  int64_t r = a.chunk_row[c * a.chunk_mul];
  int64_t rbase = r;
  auto load_row_ends = [&](int64_t rb) -> int {
    const int64_t rr = rb + 1 + lane;
    if (rr > a.n_rows) return kRowEndBeyond;
    const int64_t e = a.row_ptr[rr] - n0;
    return e > (int64_t)chunk ? chunk + 1 : (int)e;
  };
  int re_v = load_row_ends(rbase);
  int re = rl_i(re_v, 0);
  int rs;
  {
    const int64_t s0 = a.row_ptr[r] - n0;
    rs = s0 < 0 ? -1 : (int)s0;
  }

  V acc = V(0.0f);
  float *const pslot = a.partial + (size_t)c * 2 * a.pwidth + col0;

  // Sum over the slots: offsets below 16 lanes rotate inside a 16-lane row with a
  // DPP move (no LDS traffic), offsets 16 and 32 cross rows with a shuffle.
  auto xadd = [&](float v, int off) -> float {
    if (off == 1) return v + __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true));
    if (off == 2) return v + __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xF, 0xF, true));
    if (off == 4) return v + __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x124, 0xF, 0xF, true));  // row_ror:4
    if (off == 8) return v + __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x128, 0xF, 0xF, true));  // row_ror:8
    return v + __shfl_xor(v, off, kWave);
  };
  auto reduce_slots = [&](V x) -> V {
    // rotations by 8 then 4 sum the lanes {l, l+4, l+8, l+12} of a row for LW == 4;
    // in general: every power of two from LW up to 8 as a rotation, 16 and 32 as xor
#pragma unroll
    for (int off = LW; off < kWave; off <<= 1) {
      if constexpr (VEC == 1) {
        x = xadd(x, off);
      } else {
#pragma unroll
        for (int i = 0; i < VEC; ++i) x[i] = xadd(x[i], off);
      }
    }
    return x;
  };

  auto close_row = [&]() {
    const V tot = reduce_slots(acc);
    if (active && slot == 0) {
      if (rs < 0) {
        vstore<VEC>(pslot, tot);
      } else {
        const V out = epilogue_store<VEC>(a, r, col0, tot);
        if constexpr (VEC == 4 && LW >= 8) {
          // whole groups of 8 lanes are active (the host admits sign words only
          // for widths that are multiples of 32)
          if (a.sign_bits) store_sign_word(a, r, col0, lane, out);
        }
      }
    }
    acc = V(0.0f);
    rs = re;
    ++r;
    int idx = (int)(r - rbase);
    if (idx == kWave) {
      rbase = r;
      re_v = load_row_ends(rbase);
      idx = 0;
    }
    re = rl_i(re_v, idx);
  };

  auto load_batch = [&](int kb, int &cols, float &vals) {
    const int k = kb + lane;
    cols = 0;
    vals = 0.0f;
    if (k < nk) {
      const int64_t g = n0 + k;
      cols = a.col_idx[g];
      float v = a.val[g];
      if (a.vmask) {
        const int64_t mi = a.mindex ? a.mindex[g] : g;
        v = a.vmask[mi] ? v * a.mscale : 0.0f;
      }
      vals = v;
    }
  };

  int cols_n;
  float vals_n;
  load_batch(0, cols_n, vals_n);

  for (int kb = 0; kb < nk; kb += kWave) {
    const int cols = cols_n;
    const float vals = vals_n;
    if (kb + kWave < nk) load_batch(kb + kWave, cols_n, vals_n);
    const int jn = (nk - kb) < kWave ? (nk - kb) : kWave;
    for (int j = 0; j < jn; j += U * S) {
      V b[U];
      float vv[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int idx = j + u * S + slot;
        const int idc = idx < jn ? idx : (jn - 1);
        const int cc = __shfl(cols, idc, kWave);
        const float v = __shfl(vals, idc, kWave);
        vv[u] = idx < jn ? v : 0.0f;
        b[u] = V(0.0f);
        if (active)
          b[u] = vload<VEC>((cc < a.split ? a.B + (int64_t)cc * a.ldb
                                          : a.B2 + (int64_t)(cc - a.split) * a.ldb2) + col0);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (j + u * S < jn) {
          const int g0 = kb + j + u * S;                       // first nonzero of the group
          const int gend = (j + u * S + S) < jn ? (g0 + S) : (kb + jn);
          const int my_k = g0 + slot;
          const V term = (my_k < gend) ? b[u] * vv[u] : V(0.0f);
          int lo = g0;
          while (re <= gend) {                                 // rows ending inside this group
            const int e = re;
            if (my_k >= lo && my_k < e) acc += term;
            close_row();
            lo = e;
          }
          if (my_k >= lo) acc += term;
        }
      }
    }
  }
  while (re <= nk) close_row();
  if (r < a.n_rows && rs < nk) {
    const V tot = reduce_slots(acc);
    if (active && slot == 0) vstore<VEC>(pslot + (rs < 0 ? 0 : a.pwidth), tot);
  }
}

// Small matrices with a narrow dense operand (the Planetoid graphs: 10^4..10^6
// nonzeros): the chunk walk above is one long dependent chain per wavefront and
// a small matrix has too few chunks to hide it.  Here ONE wavefront owns one
// row: the S = 64 / LW slots take the row's nonzeros S at a time, the slot sums
// are added with shuffles, slot 0 applies the epilogue.  No partial sums, no
// second launch; the result is a fixed function of the row (reproducible).
template <int VEC, int LW>
__global__ __launch_bounds__(kBlock) void spmm_row_kernel(const SpmmArgs a) {
  typedef typename vec_of<VEC>::type V;
  constexpr int S = kWave / LW;
  const int lane = threadIdx.x & (kWave - 1);
  const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int64_t r = (int64_t)blockIdx.x * kWavesPerBlock + wid;
  if (r >= a.n_rows) return;
  const int slot = lane / LW;
  const int col0 = (lane % LW) * VEC;
  const bool active = col0 < a.width;
  const int64_t beg = a.row_ptr[r];
  const int64_t end = a.row_ptr[r + 1];
  V acc0 = V(0.0f), acc1 = V(0.0f);
  auto term = [&](int64_t g) -> V {
    if (g >= end || !active) return V(0.0f);
    const int cc = a.col_idx[g];
    float v = a.val[g];
    if (a.vmask) {
      const int64_t mi = a.mindex ? a.mindex[g] : g;
      v = a.vmask[mi] ? v * a.mscale : 0.0f;
    }
    const float *src = (cc < a.split ? a.B + (int64_t)cc * a.ldb
                                     : a.B2 + (int64_t)(cc - a.split) * a.ldb2) + col0;
    return vload<VEC>(src) * v;
  };
  for (int64_t k0 = beg; k0 < end; k0 += 2 * S) {
    const V t0 = term(k0 + slot);
    const V t1 = term(k0 + S + slot);
    acc0 += t0;
    acc1 += t1;
  }
  V tot = acc0 + acc1;
#pragma unroll
  for (int off = LW; off < kWave; off <<= 1) {
    if constexpr (VEC == 1) {
      tot += __shfl_xor(tot, off, kWave);
    } else {
#pragma unroll
      for (int i = 0; i < VEC; ++i) tot[i] += __shfl_xor(tot[i], off, kWave);
    }
  }
  if (active && slot == 0) epilogue_store<VEC>(a, r, col0, tot);
}

// One wavefront per chunk boundary c | c+1: if the row that owns the first
// nonzero of chunk c+1 begins inside chunk c, this wavefront completes it.
template <int VEC>
__global__ __launch_bounds__(kBlock) void spmm_fixup_kernel(const SpmmArgs a) {
  typedef typename vec_of<VEC>::type V;
  const int lane = threadIdx.x & (kWave - 1);
  const int wid = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int64_t c = (int64_t)blockIdx.x * kWavesPerBlock + wid;
  if (c + 1 >= a.n_chunks) return;
  const int col0 = (int)blockIdx.y * (kWave * VEC) + lane * VEC;
  if (col0 >= a.width) return;
  const int64_t r = a.chunk_row[(c + 1) * a.chunk_mul];
  const int64_t start = a.row_ptr[r];
  if (start >= (c + 1) * (int64_t)a.chunk || start < c * (int64_t)a.chunk) return;
  const int64_t end = a.row_ptr[r + 1];
  const int64_t cl = (end - 1) / a.chunk;
  const float *p = a.partial + col0;
  V sum = vload<VEC>(p + ((size_t)c * 2 + 1) * a.pwidth);
  for (int64_t cc = c + 1; cc <= cl; ++cc) sum += vload<VEC>(p + (size_t)cc * 2 * a.pwidth);
  const V out = epilogue_store<VEC>(a, r, col0, sum);
  if constexpr (VEC == 4) {
    if (a.sign_bits) store_sign_word(a, r, col0, lane, out);
  }
}

// The same for narrow operands (at most 32 lanes of 4 columns): a wavefront
// takes 64 / LW chunk boundaries, LW lanes each, instead of idling most lanes.
template <int LW>
__global__ __launch_bounds__(kBlock) void spmm_fixup_narrow_kernel(const SpmmArgs a) {
  typedef typename vec_of<4>::type V;
  constexpr int S = kWave / LW;
  const int lane = threadIdx.x & (kWave - 1);
  const int wid = (int)(threadIdx.x >> 6);
  const int64_t c = ((int64_t)blockIdx.x * kWavesPerBlock + wid) * S + lane / LW;
  if (c + 1 >= a.n_chunks) return;
  const int col0 = (lane % LW) * 4;
  if (col0 >= a.width) return;
  const int64_t r = a.chunk_row[(c + 1) * a.chunk_mul];
  const int64_t start = a.row_ptr[r];
  if (start >= (c + 1) * (int64_t)a.chunk || start < c * (int64_t)a.chunk) return;
  const int64_t end = a.row_ptr[r + 1];
  const int64_t cl = (end - 1) / a.chunk;
  const float *p = a.partial + col0;
  V sum = vload<4>(p + ((size_t)c * 2 + 1) * a.pwidth);
  for (int64_t cc = c + 1; cc <= cl; ++cc) sum += vload<4>(p + (size_t)cc * 2 * a.pwidth);
  const V out = epilogue_store<4>(a, r, col0, sum);
  if constexpr (LW >= 8) {
    // a boundary's lanes form whole groups of 8 (widths with sign words are multiples of 32)
    if (a.sign_bits) store_sign_word(a, r, col0, lane, out);
  }
}

// nnz == 0: every row is empty, C = act(beta * C).
__global__ void spmm_empty_kernel(const SpmmArgs a) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n_rows * (int64_t)a.width) return;
  float *dst = a.C + (i / a.width) * a.ldc + (i % a.width);
  float out = a.beta != 0.0f ? *dst * a.beta : 0.0f;
  if (a.relu && out < 0.0f) out = 0.0f;   // a NaN stays a NaN
  if (a.drop_keep < 1.0f) {
    const uint64_t dr = a.drop_rows ? (uint64_t)a.drop_rows[i / a.width] : (uint64_t)(i / a.width);
    const uint64_t e0 = a.drop_offset + dr * (uint64_t)a.drop_ld + (uint64_t)(a.drop_col0 + (i % a.width));
    const philox4 rnd = philox4x32_10(e0 >> 2, 0ull, a.drop_seed);
    out = ((float)(rnd.v[e0 & 3] >> 8) * (1.0f / 16777216.0f) < a.drop_keep) ? out / a.drop_keep : 0.0f;
  }
  *dst = out;
}

// chunk_row[c] = last row whose first slot is <= c * kChunkUnit (row 0 for
// c == 0, n_rows for c == n_chunks).
__global__ void chunk_table_kernel(int64_t n_rows, int64_t n_chunks,
                                   const int64_t *__restrict__ row_ptr,
                                   int32_t *__restrict__ chunk_row) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c > n_chunks) return;
  if (c == 0) { chunk_row[0] = 0; return; }
  if (c == n_chunks) { chunk_row[c] = (int32_t)n_rows; return; }
  const int64_t target = c * (int64_t)kChunkUnit;
  int64_t lo = 0, hi = n_rows;  // invariant: row_ptr[lo] <= target < row_ptr[hi]
  while (hi - lo > 1) {
    const int64_t mid = lo + (hi - lo) / 2;
    if (row_ptr[mid] <= target) lo = mid; else hi = mid;
  }
  chunk_row[c] = (int32_t)lo;
}

static inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace gcnx

using namespace gcnx;

static thread_local hipEvent_t g_prof_start = nullptr, g_prof_stop = nullptr;

extern "C" void gcnx_profile_spmm_events(void *start, void *stop) {
  g_prof_start = static_cast<hipEvent_t>(start);
  g_prof_stop = static_cast<hipEvent_t>(stop);
}

// Matrices with at most this many nonzeros, at most kRowPathMaxAvg per row on
// average and a dense operand of at most 32 lanes take the wavefront-per-row
// kernel (its cost is the longest row, so it is kept to small, short-rowed
// matrices: the Planetoid adjacencies and feature matrices).
constexpr int64_t kRowPathMaxNnz = (int64_t)1 << 18;
constexpr int64_t kRowPathMaxAvg = 64;

// Nonzeros per wavefront for a matrix of nnz nonzeros: the largest power of two
// in [kChunkUnit, kChunk] that still leaves about 8 wavefronts per SIMD
// (256 CUs x 4 SIMDs x 8), so that small matrices are cut finely enough to
// fill the chip and large ones amortise the per-chunk set-up.  A nonzero hint
// (gcnx_csr::spmm_chunk) overrides the choice: it is rounded up to one of
// 64, 128, 256, 512.
static int chunk_for(int64_t nnz, int32_t hint) {
  int c = kChunkUnit;
  if (hint > 0) {
    while (c < kChunk && c < hint) c *= 2;
    return c;
  }
  while (c < kChunk && nnz / (2 * c) >= 8192) c *= 2;
  return c;
}

extern "C" int gcnx_spmm_chunk_nnz(void) { return kChunkUnit; }

extern "C" int64_t gcnx_spmm_num_chunks(int64_t nnz) { return nnz <= 0 ? 0 : ceil_div(nnz, kChunkUnit); }

static void static_scratch_warm();

extern "C" int gcnx_spmm_chunk_table(const gcnx_csr *a, int32_t *chunk_row, void *stream) {
  GCNX_REQUIRE(a && chunk_row, GCNX_ERR_INVALID_ARG, "chunk_table: null argument");
  static_scratch_warm();
  GCNX_REQUIRE(a->n_rows >= 0 && a->nnz >= 0 && a->n_rows < INT32_MAX, GCNX_ERR_INVALID_ARG,
               "chunk_table: bad shape");
  const int64_t n_chunks = gcnx_spmm_num_chunks(a->nnz);
  const int64_t threads = n_chunks + 1;
  hipLaunchKernelGGL(chunk_table_kernel, dim3((unsigned)ceil_div(threads, 256)), dim3(256), 0,
                     as_stream(stream), a->n_rows, n_chunks, a->row_ptr, chunk_row);
  GCNX_CHECK_LAUNCH();
  return GCNX_OK;
}

static inline int32_t partial_ld(int32_t width) { return (width + 3) / 4 * 4; }

static size_t partial_bytes(int64_t nnz, int32_t hint, int32_t width) {
  if (nnz <= 0 || width <= 0) return 0;
  return (size_t)ceil_div(nnz, chunk_for(nnz, hint)) * 2 * (size_t)partial_ld(width) * sizeof(float);
}

extern "C" size_t gcnx_spmm_workspace_bytes(const gcnx_csr *a, int32_t width) {
  return a ? partial_bytes(a->nnz, a->spmm_chunk, width) : 0;
}

extern "C" int gcnx_spmm_csr_f32(const gcnx_csr *a, const int32_t *chunk_row, const float *B,
                                 int64_t ldb, float *C, int64_t ldc, int32_t width, float alpha,
                                 float beta, const uint8_t *val_mask, const int64_t *mask_index,
                                 float mask_scale, int fuse_relu, void *ws, size_t ws_bytes,
                                 void *stream) {
  return gcnx_spmm_csr_split_f32(a, chunk_row, B, ldb, nullptr, 0, a ? a->n_cols : 0, C, ldc, width,
                                 alpha, beta, val_mask, mask_index, mask_scale, fuse_relu, nullptr,
                                 ws, ws_bytes, stream);
}

static int spmm_impl(const gcnx_csr *a, const int32_t *chunk_row, const float *B, int64_t ldb,
                     const float *B2, int64_t ldb2, int64_t split_col, float *C, int64_t ldc,
                     int32_t width, float alpha, float beta, const uint8_t *val_mask,
                     const int64_t *mask_index, float mask_scale, int fuse_relu,
                     const gcnx_dropout *out_dropout, const int32_t *drop_rows, uint32_t *sign_bits,
                     int64_t ld_sign, void *ws, size_t ws_bytes, void *stream);

extern "C" int gcnx_spmm_csr_split_f32(const gcnx_csr *a, const int32_t *chunk_row, const float *B,
                                       int64_t ldb, const float *B2, int64_t ldb2,
                                       int64_t split_col, float *C, int64_t ldc, int32_t width,
                                       float alpha, float beta, const uint8_t *val_mask,
                                       const int64_t *mask_index, float mask_scale, int fuse_relu,
                                       const gcnx_dropout *out_dropout, void *ws, size_t ws_bytes,
                                       void *stream) {
  return spmm_impl(a, chunk_row, B, ldb, B2, ldb2, split_col, C, ldc, width, alpha, beta, val_mask, mask_index,
                   mask_scale, fuse_relu, out_dropout, nullptr, nullptr, 0, ws, ws_bytes, stream);
}

extern "C" int gcnx_spmm_signs_supported(int32_t width) { return (width >= 32 && width % 32 == 0) ? 1 : 0; }

extern "C" int gcnx_spmm_csr_signs_f32(const gcnx_csr *a, const int32_t *chunk_row, const float *B,
                                       int64_t ldb, const float *B2, int64_t ldb2, int64_t split_col,
                                       float *C, int64_t ldc, int32_t width, float alpha, float beta,
                                       int fuse_relu, const gcnx_dropout *out_dropout,
                                       uint32_t *sign_bits, int64_t ld_words, void *ws, size_t ws_bytes,
                                       void *stream) {
  GCNX_REQUIRE(sign_bits && ld_words * 32 >= width && gcnx_spmm_signs_supported(width), GCNX_ERR_UNSUPPORTED,
               "spmm_signs: needs width %% 32 == 0 and ld_words * 32 >= width");
  return spmm_impl(a, chunk_row, B, ldb, B2, ldb2, split_col, C, ldc, width, alpha, beta, nullptr, nullptr,
                   1.0f, fuse_relu, out_dropout, nullptr, sign_bits, ld_words, ws, ws_bytes, stream);
}

extern "C" int gcnx_spmm_csr_rows_f32(const gcnx_csr *a, const int32_t *chunk_row, const float *B,
                                      int64_t ldb, const float *B2, int64_t ldb2, int64_t split_col,
                                      float *C, int64_t ldc, int32_t width, float alpha, float beta,
                                      int fuse_relu, const gcnx_dropout *out_dropout,
                                      const int32_t *dropout_row_ids, uint32_t *sign_bits, int64_t ld_words,
                                      void *ws, size_t ws_bytes, void *stream) {
  GCNX_REQUIRE(!sign_bits || (ld_words * 32 >= width && gcnx_spmm_signs_supported(width)), GCNX_ERR_UNSUPPORTED,
               "spmm_rows: sign words need width %% 32 == 0 and ld_words * 32 >= width");
  return spmm_impl(a, chunk_row, B, ldb, B2, ldb2, split_col, C, ldc, width, alpha, beta, nullptr, nullptr,
                   1.0f, fuse_relu, out_dropout, dropout_row_ids, sign_bits, ld_words, ws, ws_bytes, stream);
}

static int spmm_impl(const gcnx_csr *a, const int32_t *chunk_row, const float *B, int64_t ldb,
                     const float *B2, int64_t ldb2, int64_t split_col, float *C, int64_t ldc,
                     int32_t width, float alpha, float beta, const uint8_t *val_mask,
                     const int64_t *mask_index, float mask_scale, int fuse_relu,
                     const gcnx_dropout *out_dropout, const int32_t *drop_rows, uint32_t *sign_bits,
                     int64_t ld_sign, void *ws, size_t ws_bytes, void *stream) {
  GCNX_REQUIRE(a, GCNX_ERR_INVALID_ARG, "spmm: null matrix");
  GCNX_REQUIRE(a->n_rows >= 0 && a->n_cols >= 0 && a->nnz >= 0 && width >= 0,
               GCNX_ERR_INVALID_ARG, "spmm: negative size");
  GCNX_REQUIRE(a->n_rows < INT32_MAX && a->n_cols < INT32_MAX, GCNX_ERR_UNSUPPORTED,
               "spmm: more than 2^31-1 rows or columns");
  if (a->n_rows == 0 || width == 0) return GCNX_OK;   // an empty output may come with a null pointer
  GCNX_REQUIRE(C, GCNX_ERR_INVALID_ARG, "spmm: null output");
  GCNX_REQUIRE(ldc >= width, GCNX_ERR_INVALID_ARG, "spmm: ldc < width");
  hipStream_t st = as_stream(stream);

  SpmmArgs k;
  k.n_rows = a->n_rows;
  k.nnz = a->nnz;
  k.chunk = chunk_for(a->nnz, a->spmm_chunk);
  k.chunk_mul = k.chunk / kChunkUnit;
  k.n_chunks = a->nnz <= 0 ? 0 : ceil_div(a->nnz, k.chunk);
  k.row_ptr = a->row_ptr;
  k.col_idx = a->col_idx;
  k.val = a->val;
  k.chunk_row = chunk_row;
  k.keep_bits = reinterpret_cast<const unsigned long long *>(a->gather_keep);
  k.B = B;
  k.ldb = ldb;
  k.B2 = B2;
  k.ldb2 = ldb2;
  k.split = INT32_MAX;
  k.C = C;
  k.ldc = ldc;
  k.width = width;
  k.pwidth = partial_ld(width);
  k.alpha = alpha;
  k.beta = beta;
  k.vmask = val_mask;
  k.mindex = mask_index;
  k.mscale = mask_scale;
  k.relu = fuse_relu;
  k.partial = static_cast<float *>(ws);
  k.drop_keep = 1.0f;
  k.drop_seed = 0;
  k.drop_offset = 0;
  k.drop_ld = 0;
  k.drop_col0 = 0;
  k.drop_rows = nullptr;
  k.sign_bits = sign_bits;
  k.ld_sign = ld_sign;
  bool drop_vec_ok = true;
  if (out_dropout && out_dropout->keep < 1.0f) {
    GCNX_REQUIRE(out_dropout->keep > 0.0f && out_dropout->ld >= width && out_dropout->col0 >= 0,
                 GCNX_ERR_INVALID_ARG, "spmm: bad output dropout descriptor");
    k.drop_keep = out_dropout->keep;
    k.drop_seed = out_dropout->seed;
    k.drop_offset = out_dropout->offset;
    k.drop_ld = out_dropout->ld;
    k.drop_col0 = out_dropout->col0;
    k.drop_rows = drop_rows;
    drop_vec_ok = (out_dropout->offset % 4 == 0) && (out_dropout->ld % 4 == 0) && (out_dropout->col0 % 4 == 0);
  }

  GCNX_REQUIRE(!sign_bits || a->nnz > 0, GCNX_ERR_UNSUPPORTED, "spmm_signs: matrix without nonzeros");
  if (a->nnz == 0) {
    const int64_t total = a->n_rows * (int64_t)width;
    hipLaunchKernelGGL(spmm_empty_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, st, k);
    GCNX_CHECK_LAUNCH();
    return GCNX_OK;
  }
  GCNX_REQUIRE(a->row_ptr && a->col_idx && a->val && chunk_row && (B || split_col == 0),
               GCNX_ERR_INVALID_ARG, "spmm: null array");
  GCNX_REQUIRE(ldb >= width, GCNX_ERR_INVALID_ARG, "spmm: ldb < width");
  if (split_col < a->n_cols) {
    GCNX_REQUIRE(split_col >= 0 && B2 && ldb2 >= width, GCNX_ERR_INVALID_ARG,
                 "spmm: second operand missing or ldb2 < width");
    k.split = (int32_t)split_col;
  }
  const size_t need = gcnx_spmm_workspace_bytes(a, width);
  GCNX_REQUIRE(ws && ws_bytes >= need && aligned16(ws), GCNX_ERR_WORKSPACE,
               "spmm: workspace of %zu bytes (16-byte aligned) required, got %zu", need, ws_bytes);
  GCNX_REQUIRE(k.n_chunks / kWavesPerBlock < INT32_MAX, GCNX_ERR_UNSUPPORTED, "spmm: too many chunks");

  const bool v4 = (width % 4 == 0) && (ldb % 4 == 0) && (ldc % 4 == 0) && aligned16(B) && aligned16(C) &&
                  (k.split == INT32_MAX || ((ldb2 % 4 == 0) && aligned16(B2))) && drop_vec_ok;
  const unsigned gx = (unsigned)ceil_div(k.n_chunks, kWavesPerBlock);
  const int vec = v4 ? 4 : 1;
  const int lanes = (width + vec - 1) / vec;      // lanes one row of B needs
  unsigned gy = 1;
  if (g_prof_start) GCNX_CHECK_HIP(hipEventRecord(g_prof_start, st));
  if (!sign_bits && lanes <= 32 && a->spmm_chunk <= 0 && a->nnz <= kRowPathMaxNnz &&
      a->nnz <= kRowPathMaxAvg * a->n_rows) {
    const unsigned gr = (unsigned)ceil_div(a->n_rows, kWavesPerBlock);
#define GCNX_ROW(V, L) hipLaunchKernelGGL((spmm_row_kernel<V, L>), dim3(gr), dim3(kBlock), 0, st, k)
    if (v4) {
      if (lanes > 16) GCNX_ROW(4, 32);
      else if (lanes > 8) GCNX_ROW(4, 16);
      else if (lanes > 4) GCNX_ROW(4, 8);
      else if (lanes > 2) GCNX_ROW(4, 4);
      else if (lanes > 1) GCNX_ROW(4, 2);
      else GCNX_ROW(4, 1);
    } else {
      if (lanes > 16) GCNX_ROW(1, 32);
      else if (lanes > 8) GCNX_ROW(1, 16);
      else if (lanes > 4) GCNX_ROW(1, 8);
      else if (lanes > 2) GCNX_ROW(1, 4);
      else if (lanes > 1) GCNX_ROW(1, 2);
      else GCNX_ROW(1, 1);
    }
#undef GCNX_ROW
    GCNX_CHECK_LAUNCH();
    if (g_prof_stop) GCNX_CHECK_HIP(hipEventRecord(g_prof_stop, st));
    return GCNX_OK;
  }
#ifndef GCNX_SLOT_U
#define GCNX_SLOT_U 4       // gather groups in flight per wavefront in the narrow kernel
#endif
#define GCNX_SLOT(V, L) hipLaunchKernelGGL((spmm_slot_kernel<V, L, GCNX_SLOT_U>), dim3(gx), dim3(kBlock), 0, st, k)
  GCNX_REQUIRE(!sign_bits || (v4 && width % 32 == 0), GCNX_ERR_UNSUPPORTED,
               "spmm_signs: operands must be 16-byte aligned with rows that are multiples of 4 floats");
  if (lanes > 32) {
    gy = (unsigned)ceil_div(lanes, kWave);
    if (v4) hipLaunchKernelGGL((spmm_chunk_kernel<4, GCNX_WIDE_U>), dim3(gx, gy), dim3(kBlock), 0, st, k);
    else hipLaunchKernelGGL((spmm_chunk_kernel<1, GCNX_WIDE_U>), dim3(gx, gy), dim3(kBlock), 0, st, k);
  } else if (v4) {
    if (lanes > 16) GCNX_SLOT(4, 32);
    else if (lanes > 8) GCNX_SLOT(4, 16);
    else if (lanes > 4) GCNX_SLOT(4, 8);
    else if (lanes > 2) GCNX_SLOT(4, 4);
    else if (lanes > 1) GCNX_SLOT(4, 2);
    else GCNX_SLOT(4, 1);
  } else {
    if (lanes > 16) GCNX_SLOT(1, 32);
    else if (lanes > 8) GCNX_SLOT(1, 16);
    else if (lanes > 4) GCNX_SLOT(1, 8);
    else if (lanes > 2) GCNX_SLOT(1, 4);
    else if (lanes > 1) GCNX_SLOT(1, 2);
    else GCNX_SLOT(1, 1);
  }
#undef GCNX_SLOT
  GCNX_CHECK_LAUNCH();
  if (k.n_chunks > 1) {
    if (v4 && lanes <= 32) {
      // narrow operand: several boundaries per wavefront
#define GCNX_FIXN(L) hipLaunchKernelGGL((spmm_fixup_narrow_kernel<L>), \
    dim3((unsigned)ceil_div(k.n_chunks, (int64_t)kWavesPerBlock * (kWave / L))), dim3(kBlock), 0, st, k)
      if (lanes > 16) GCNX_FIXN(32);
      else if (lanes > 8) GCNX_FIXN(16);
      else if (lanes > 4) GCNX_FIXN(8);
      else GCNX_FIXN(4);
#undef GCNX_FIXN
    } else if (v4) hipLaunchKernelGGL((spmm_fixup_kernel<4>), dim3(gx, gy), dim3(kBlock), 0, st, k);
    else hipLaunchKernelGGL((spmm_fixup_kernel<1>), dim3(gx, gy), dim3(kBlock), 0, st, k);
    GCNX_CHECK_LAUNCH();
  }
  // the timed window covers the whole product: main pass and fix-up pass
  if (g_prof_stop) GCNX_CHECK_HIP(hipEventRecord(g_prof_stop, st));
  return GCNX_OK;
}

// ---- the product for callers that cannot allocate --------------------------
// The partial sums of rows cut by chunk boundaries go to a file-scope device
// array instead of a caller workspace.  Every GPU the library runs on holds its
// own copy of the array at its own address until the process ends; the address
// is looked up once per device (hipGetSymbolAddress with that device current)
// and only that pointer is ever handed to kernels -- the array's name in host
// code is a host-side stand-in, not a device address.
static __device__ __align__(256) unsigned char scratch[GCNX_SPMM_STATIC_SCRATCH_BYTES];

namespace {
constexpr int kMaxDevices = 64;
std::mutex g_scratch_mu;
// per device; published once (under the mutex) and read without it afterwards
std::atomic<void *> g_scratch_ptr[kMaxDevices];

// Device address of `scratch` on the current device.  The first call on a device
// looks it up (hipGetSymbolAddress loads the code object and maps the array: the
// one moment the runtime may allocate, so it must not fall into a stream
// capture -- gcnx_spmm_chunk_table, which every matrix needs before its first
// product and which no caller runs under capture, makes that call); every later
// call is one atomic load.
int static_scratch(void **out) {
  int dev = -1;
  GCNX_CHECK_HIP(hipGetDevice(&dev));
  GCNX_REQUIRE(dev >= 0 && dev < kMaxDevices, GCNX_ERR_UNSUPPORTED, "spmm_static: device ordinal %d out of range", dev);
  void *p = g_scratch_ptr[dev].load(std::memory_order_acquire);
  if (!p) {
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    p = g_scratch_ptr[dev].load(std::memory_order_relaxed);
    if (!p) {
      GCNX_CHECK_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
      GCNX_REQUIRE(p != nullptr, GCNX_ERR_HIP, "spmm_static: the scratch array has no address on device %d", dev);
      g_scratch_ptr[dev].store(p, std::memory_order_release);
    }
  }
  *out = p;
  return GCNX_OK;
}
}  // namespace

// Called by gcnx_spmm_chunk_table (set-up, never under capture): resolves the
// scratch array's address on the current device so that the no-allocation entry
// point finds it published.  A failure here is not the chunk table's: it is
// reported by gcnx_spmm_csr_static_f32 itself if that is ever called.
static void static_scratch_warm() {
  void *p = nullptr;
  (void)static_scratch(&p);
}

extern "C" int gcnx_spmm_csr_static_f32(const gcnx_csr *a, const int32_t *chunk_row, const float *B,
                                        int64_t ldb, float *C, int64_t ldc, int32_t width, float alpha,
                                        float beta, const uint8_t *val_mask, const int64_t *mask_index,
                                        float mask_scale, int fuse_relu, void *stream) {
  GCNX_REQUIRE(a, GCNX_ERR_INVALID_ARG, "spmm_static: null matrix");
  const size_t need = gcnx_spmm_workspace_bytes(a, width);
  GCNX_REQUIRE(need <= (size_t)GCNX_SPMM_STATIC_SCRATCH_BYTES, GCNX_ERR_WORKSPACE,
               "spmm_static: this product needs %zu bytes of partial sums, the built-in scratch array holds %zu; "
               "use gcnx_spmm_csr_f32 with a caller workspace",
               need, (size_t)GCNX_SPMM_STATIC_SCRATCH_BYTES);
  void *ws = nullptr;
  const int rc = static_scratch(&ws);
  if (rc != GCNX_OK) return rc;
  return spmm_impl(a, chunk_row, B, ldb, nullptr, 0, a->n_cols, C, ldc, width, alpha, beta, val_mask, mask_index,
                   mask_scale, fuse_relu, nullptr, nullptr, nullptr, 0, ws, (size_t)GCNX_SPMM_STATIC_SCRATCH_BYTES, stream);
}

#ifndef GCNX_FACTORED_U
#define GCNX_FACTORED_U 3   // nonzeros whose operand rows are in flight per wavefront (16 scalar registers each)
#endif

extern "C" int gcnx_spmm_factored_supported(int32_t width, int32_t inner) {
  return (inner == kFactoredInner && width >= 64 && width % 32 == 0) ? 1 : 0;
}

extern "C" int gcnx_spmm_csr_factored_f32(const gcnx_csr *a, const int32_t *chunk_row, const float *D,
                                          int64_t ldd, const uint32_t *gate_bits, int64_t ld_words,
                                          const float *D2, int64_t ldd2, const uint32_t *gate_bits2,
                                          int64_t ld_words2, int64_t split_col, const float *W, int64_t ldw,
                                          int trans_w, int32_t inner, float gate_scale, float *C,
                                          int64_t ldc, int32_t width, void *ws, size_t ws_bytes,
                                          void *stream) {
  GCNX_REQUIRE(a, GCNX_ERR_INVALID_ARG, "spmm_factored: null matrix");
  GCNX_REQUIRE(a->n_rows >= 0 && a->n_cols >= 0 && a->nnz >= 0 && width >= 0, GCNX_ERR_INVALID_ARG,
               "spmm_factored: negative size");
  GCNX_REQUIRE(a->n_rows < INT32_MAX && a->n_cols < INT32_MAX, GCNX_ERR_UNSUPPORTED,
               "spmm_factored: more than 2^31-1 rows or columns");
  GCNX_REQUIRE(gcnx_spmm_factored_supported(width, inner), GCNX_ERR_UNSUPPORTED,
               "spmm_factored: needs inner == %d, width >= 64 and width %% 32 == 0", kFactoredInner);
  if (a->n_rows == 0) return GCNX_OK;
  GCNX_REQUIRE(C && ldc >= width && (ldc % 4 == 0) && aligned16(C), GCNX_ERR_INVALID_ARG,
               "spmm_factored: C must be 16-byte aligned with ldc >= width, ldc %% 4 == 0");
  GCNX_REQUIRE(W && ldw >= (trans_w ? inner : width), GCNX_ERR_INVALID_ARG, "spmm_factored: bad W");
  hipStream_t st = as_stream(stream);
  SpmmArgs k = {};
  k.n_rows = a->n_rows;
  k.nnz = a->nnz;
  k.chunk = chunk_for(a->nnz, a->spmm_chunk);
  k.chunk_mul = k.chunk / kChunkUnit;
  k.n_chunks = a->nnz <= 0 ? 0 : ceil_div(a->nnz, k.chunk);
  k.row_ptr = a->row_ptr;
  k.col_idx = a->col_idx;
  k.val = a->val;
  k.chunk_row = chunk_row;
  k.split = INT32_MAX;
  k.C = C;
  k.ldc = ldc;
  k.width = width;
  k.pwidth = partial_ld(width);
  k.alpha = 1.0f;
  k.mscale = 1.0f;
  k.partial = static_cast<float *>(ws);
  k.drop_keep = 1.0f;
  if (a->nnz == 0) {   // every row empty: C = 0
    hipLaunchKernelGGL(spmm_empty_kernel, dim3((unsigned)ceil_div(a->n_rows * (int64_t)width, 256)), dim3(256), 0,
                       st, k);
    GCNX_CHECK_LAUNCH();
    return GCNX_OK;
  }
  GCNX_REQUIRE(a->row_ptr && a->col_idx && a->val && chunk_row, GCNX_ERR_INVALID_ARG, "spmm_factored: null array");
  GCNX_REQUIRE(split_col >= 0 && (split_col == 0 || (D && gate_bits && ldd >= inner && ld_words * 32 >= width)),
               GCNX_ERR_INVALID_ARG, "spmm_factored: bad D / gate words");
  GCNX_REQUIRE((reinterpret_cast<uintptr_t>(D) & 3) == 0 && (reinterpret_cast<uintptr_t>(D2) & 3) == 0,
               GCNX_ERR_INVALID_ARG, "spmm_factored: D must be 4-byte aligned");
  if (split_col < a->n_cols) {
    GCNX_REQUIRE(D2 && gate_bits2 && ldd2 >= inner && ld_words2 * 32 >= width, GCNX_ERR_INVALID_ARG,
                 "spmm_factored: second operand missing or too narrow");
    k.split = (int32_t)split_col;
  }
  const size_t need = gcnx_spmm_workspace_bytes(a, width);
  GCNX_REQUIRE(ws && ws_bytes >= need && aligned16(ws), GCNX_ERR_WORKSPACE,
               "spmm_factored: workspace of %zu bytes (16-byte aligned) required, got %zu", need, ws_bytes);
  GCNX_REQUIRE(k.n_chunks / kWavesPerBlock < INT32_MAX, GCNX_ERR_UNSUPPORTED, "spmm_factored: too many chunks");
  k.D = D; k.ldd = ldd; k.D2 = D2; k.ldd2 = ldd2;
  k.gbits = gate_bits; k.ldgb = ld_words; k.gbits2 = gate_bits2; k.ldgb2 = ld_words2;
  k.Wm = W;
  k.w_sk = trans_w ? 1 : ldw;
  k.w_sj = trans_w ? ldw : 1;
  k.fscale = gate_scale;
  const unsigned gx = (unsigned)ceil_div(k.n_chunks, kWavesPerBlock);
  const unsigned gy = (unsigned)ceil_div(width, kWave * 4);
  if (g_prof_start) GCNX_CHECK_HIP(hipEventRecord(g_prof_start, st));
  // row-forming on the matrix pipe when the operands allow 16-byte loads
  const bool mfma_ok = (ldd % 4 == 0) && aligned16(D) && (ld_words % 4 == 0) && aligned16(gate_bits) &&
                       (k.split == INT32_MAX || ((ldd2 % 4 == 0) && aligned16(D2) && (ld_words2 % 4 == 0) &&
                                                 aligned16(gate_bits2)));
  if (mfma_ok) {
    // two chunks per wavefront; 256-column slabs (8 accumulator tiles) when the
    // width fills them, 128-column slabs otherwise
    const int64_t pairs = ceil_div(k.n_chunks, 2);
    const unsigned gxp = (unsigned)ceil_div(pairs, kWavesPerBlock);
    const bool split = dense_arith_mode() == 0;
#define GCNX_FACT(NTV, FULLV, GY)                                                                                 \
  do {                                                                                                            \
    if (split) hipLaunchKernelGGL((spmm_factored_mfma_kernel<NTV, FULLV, true>), dim3(gxp, (unsigned)(GY)), dim3(kBlock), 0, st, k); \
    else hipLaunchKernelGGL((spmm_factored_mfma_kernel<NTV, FULLV, false>), dim3(gxp, (unsigned)(GY)), dim3(kBlock), 0, st, k);      \
  } while (0)
    if (width % 256 == 0) GCNX_FACT(8, true, width / 256);
    else if (width % 128 == 0) GCNX_FACT(4, true, width / 128);
    else GCNX_FACT(4, false, ceil_div(width, 128));
#undef GCNX_FACT
  } else {
    hipLaunchKernelGGL((spmm_factored_kernel<GCNX_FACTORED_U>), dim3(gx, gy), dim3(kBlock), 0, st, k);
  }
  GCNX_CHECK_LAUNCH();
  if (k.n_chunks > 1) {
    hipLaunchKernelGGL((spmm_fixup_kernel<4>), dim3(gx, gy), dim3(kBlock), 0, st, k);
    GCNX_CHECK_LAUNCH();
  }
  if (g_prof_stop) GCNX_CHECK_HIP(hipEventRecord(g_prof_stop, st));
  return GCNX_OK;
}



