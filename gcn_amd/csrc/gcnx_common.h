// Shared host-side helpers for the gcnx translation units (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/gcnx.h"

namespace gcnx {

// Thread-local error text behind gcnx_last_error().
void set_error(const char *fmt, ...);
const char *get_error();

// Arithmetic of the large dense products on this thread (gcnx_dense_arith).
int dense_arith_mode();

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

constexpr int kWave = 64;            // gfx950 wavefront
constexpr int kBlock = 256;          // 4 wavefronts per workgroup
constexpr int kWavesPerBlock = kBlock / kWave;
constexpr int kChunk = 512;          // most nonzeros one wavefront owns in SpMM
constexpr int kChunkUnit = 64;       // granularity of the chunk table (and the smallest chunk)

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

}  // namespace gcnx

#define GCNX_CHECK_HIP(expr)                                                   \
  do {                                                                         \
    hipError_t e_ = (expr);                                                    \
    if (e_ != hipSuccess) {                                                    \
      gcnx::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),   \
                      __FILE__, __LINE__);                                     \
      return GCNX_ERR_HIP;                                                     \
    }                                                                          \
  } while (0)

#define GCNX_CHECK_LAUNCH()                                                    \
  do {                                                                         \
    hipError_t e_ = hipGetLastError();                                         \
    if (e_ != hipSuccess) {                                                    \
      gcnx::set_error("kernel launch failed: %s (%s:%d)",                      \
                      hipGetErrorString(e_), __FILE__, __LINE__);              \
      return GCNX_ERR_HIP;                                                     \
    }                                                                          \
  } while (0)

#define GCNX_REQUIRE(cond, code, ...)                                          \
  do {                                                                         \
    if (!(cond)) {                                                             \
      gcnx::set_error(__VA_ARGS__);                                            \
      return code;                                                             \
    }                                                                          \
  } while (0)
