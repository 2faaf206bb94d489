/*
 * gcnx -- C ABI of the MI355X (gfx950) GCN propagation kernels.
 *
 * The reference (harleyiscoding/gcn, a TensorFlow-1.x Kipf GCN) has no FFI of
 * its own: its propagation path is a handful of TensorFlow ops reached from
 * gcn/layers.py.  Each entry point below replaces one of those call sites; the
 * citation after "replaces:" is the reference file:line (relative to
 * gcn-master/).  The host-side mirror of the reference's Python API
 * (gcn_amd/layers.py, models.py, utils.py) calls these through ctypes.
 *
 * Conventions
 *   - Every function returns GCNX_OK (0) or a negative gcnx_status; it never
 *     throws, aborts or frees caller memory.  gcnx_last_error() returns a
 *     thread-local description of the last failure.
 *   - All array arguments are DEVICE pointers borrowed from the caller, unless
 *     the name starts with h_.  The caller keeps them alive until the stream
 *     has been synchronised.
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).
 *     Compute entry points are asynchronous and stream-ordered and never
 *     allocate; scratch memory comes in through (ws, ws_bytes), sized by the
 *     matching *_workspace_bytes query.  Builder entry points (csr_from_coo,
 *     csr_transpose, rmat_*) synchronise the stream once before returning so
 *     that they can report malformed input.
 *   - Row-major dense matrices with an explicit leading dimension in elements.
 *   - Indices: row_ptr is int64 (nnz may exceed 2^31), col_idx int32.
 */
#ifndef GCNX_H
#define GCNX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GCNX_ABI_VERSION 5

typedef enum gcnx_status {
  GCNX_OK = 0,
  GCNX_ERR_INVALID_ARG = -1,  /* null pointer, negative size, bad stride ... */
  GCNX_ERR_WORKSPACE = -2,    /* ws_bytes smaller than the query's answer    */
  GCNX_ERR_INDEX_RANGE = -3,  /* a coordinate outside [0,n_rows) x [0,n_cols) */
  GCNX_ERR_HIP = -4,          /* a HIP runtime call failed                   */
  GCNX_ERR_UNSUPPORTED = -5,
  GCNX_ERR_COMM = -6          /* an RCCL call failed                         */
} gcnx_status;

/* CSR matrix resident on the device.  Entries of a row are ordered by column,
 * duplicates kept in input order.
 * spmm_chunk is a per-matrix dispatch hint read by the SpMM entry points and by
 * gcnx_spmm_workspace_bytes: 0 (the default of a zero-initialised struct) lets
 * the library choose (one wavefront per row for small short-rowed matrices,
 * otherwise chunks of 64..512 nonzeros per wavefront by matrix size); 64, 128,
 * 256 or 512 sends every product with this matrix down the chunk kernels with
 * that many nonzeros per wavefront (other values are rounded up to the next of
 * these).  The library keeps no dispatch state of its own.
 * gather_keep (nullable) is a cache hint for products whose dense operand is
 * wider than 128 columns and far larger than the caches: bit k % 64 of word
 * k / 64 says that the operand row of nonzero k is worth keeping cached (its
 * column is referenced often); rows of nonzeros whose bit is clear are read
 * with streaming (non-temporal) loads, so that they do not evict the rows that
 * are.  It changes which cache lines survive, never a result.  Build it with
 * gcnx_csr_column_degrees + gcnx_csr_gather_keep. */
typedef struct gcnx_csr {
  int64_t n_rows;
  int64_t n_cols;
  int64_t nnz;
  const int64_t *row_ptr; /* [n_rows + 1] */
  const int32_t *col_idx; /* [nnz] */
  const float *val;       /* [nnz] */
  int32_t spmm_chunk;     /* dispatch hint, see above; 0 = chosen by the library */
  const uint64_t *gather_keep; /* [ceil(nnz / 64)] or NULL, see above */
} gcnx_csr;

/* Dropout fused into the epilogue of a product (replaces tf.nn.dropout on the
 * product's output, gcn/layers.py:210): output element (r, c) is kept iff the
 * Philox uniform of (seed, offset + r * ld + col0 + c) is < keep (the rule of
 * gcnx_bernoulli_mask_u8), survivors are scaled by 1/keep.  ld / col0 locate a
 * column slab inside the full activation. */
typedef struct gcnx_dropout {
  float keep;
  uint64_t seed;
  uint64_t offset;
  int64_t ld;
  int64_t col0;
} gcnx_dropout;

int gcnx_abi_version(void);
const char *gcnx_last_error(void);
/* Granularity of the chunk table below, in nonzeros.  One wavefront of the SpMM
 * kernels owns 1, 2, 4 or 8 such units, chosen from the matrix size at each
 * call. */
int gcnx_spmm_chunk_nnz(void);

/* ---- CSR construction ---------------------------------------------------
 * replaces: the sparse tuple -> tf.sparse_placeholder(tf.float32) feed,
 * gcn/utils.py:93-109,143-144 and gcn/train_normal.py:50-51 (TensorFlow takes
 * the unsorted COO tuple on every step; here it is turned into a resident CSR
 * once).  Stable order by (row, col); duplicates kept; fp64 values rounded to
 * fp32 (nearest-even) after ordering.  Coordinates are read as
 * row[k*coord_stride], col[k*coord_stride], so an interleaved [nnz,2] int32
 * array is (coords, coords+1, 2).  perm_out[slot] = input position (nullable).
 */
size_t gcnx_csr_from_coo_workspace_bytes(int64_t n_rows, int64_t nnz);
int gcnx_csr_from_coo(int64_t n_rows, int64_t n_cols, int64_t nnz,
                      const int32_t *row, const int32_t *col,
                      int64_t coord_stride, const void *val, int val_is_f64,
                      int64_t *row_ptr, int32_t *col_idx, float *val_out,
                      int64_t *perm_out, void *ws, size_t ws_bytes,
                      void *stream);

/* Exact structural transpose (no arithmetic on the values).
 * replaces: adjoint_a=True of TensorFlow's SparseTensorDenseMatMul gradient,
 * created by optimizer.minimize at gcn/models.py:58 (backward of
 * gcn/layers.py:219 and :215).  perm_out[slot of A^T] = slot of A (nullable). */
size_t gcnx_csr_transpose_workspace_bytes(int64_t n_rows, int64_t n_cols,
                                          int64_t nnz);
int gcnx_csr_transpose(const gcnx_csr *a, int64_t *t_row_ptr,
                       int32_t *t_col_idx, float *t_val, int64_t *perm_out,
                       void *ws, size_t ws_bytes, void *stream);

/* Chunk table for the SpMM kernels: chunk_row[c] = row that owns nonzero
 * c * gcnx_spmm_chunk_nnz().  n_chunks = ceil(nnz / chunk).  Built once per
 * matrix.  chunk_row has n_chunks + 1 entries (the last is n_rows). */
int64_t gcnx_spmm_num_chunks(int64_t nnz);
int gcnx_spmm_chunk_table(const gcnx_csr *a, int32_t *chunk_row, void *stream);

/* ---- SpMM:  C = act(alpha * A' * B + beta * C) ----------------------------
 * replaces: dot(x, y, sparse=True) = tf.sparse_tensor_dense_matmul,
 * gcn/layers.py:30-33, at its call sites :219 (A_hat * P), :215-216 and :123
 * (X * W), and, given the transposed CSR, the backward products A_hat^T * dH and
 * X^T * dZ (gcn/models.py:58).  A' is A with the sparse-dropout mask applied on
 * the fly (replaces sparse_dropout, gcn/layers.py:21-27): with val_mask != NULL
 * the value of slot k is  val_mask[mask_index ? mask_index[k] : k] ?
 * val[k] * mask_scale : 0.  fuse_relu != 0 applies max(0, .) before the store
 * (gcn/layers.py:226).  beta == 0 never reads C.
 * Sums are formed in fp32 in slot order inside a chunk; rows that span several
 * chunks are completed from per-chunk partial sums in chunk order, so results
 * are bitwise reproducible run to run.
 * Matrices of at most 2^18 nonzeros (and at most 64 per row on average) whose
 * dense operand fits 32 lanes (128 columns with 16-byte aligned rows, else 32)
 * are summed one wavefront per row instead (slot-strided terms, then a shuffle
 * tree), unless a->spmm_chunk asks for the chunk kernels: also a fixed function
 * of the input.
 * Workspace: sized for the matrix (only a->nnz and a->spmm_chunk are read) and
 * the operand width; pass the same spmm_chunk to the query and to the call.
 */
size_t gcnx_spmm_workspace_bytes(const gcnx_csr *a, int32_t width);
int gcnx_spmm_csr_f32(const gcnx_csr *a, const int32_t *chunk_row,
                      const float *B, int64_t ldb, float *C, int64_t ldc,
                      int32_t width, float alpha, float beta,
                      const uint8_t *val_mask, const int64_t *mask_index,
                      float mask_scale, int fuse_relu, void *ws,
                      size_t ws_bytes, void *stream);

/* gcnx_spmm_csr_f32 for callers that cannot allocate or carry a workspace
 * (code under HIP graph capture, tight serving loops, harnesses that forbid
 * hipMalloc).  No counterpart in the reference.  The partial sums of rows cut by
 * chunk boundaries go to a device array built into the library
 * (GCNX_SPMM_STATIC_SCRATCH_BYTES; every GPU the library runs on holds one until
 * the process ends) instead of (ws, ws_bytes); the call makes no hipMalloc,
 * hipFree or other allocating call.  Products whose
 * gcnx_spmm_workspace_bytes(a, width) exceeds the array are rejected with
 * GCNX_ERR_WORKSPACE (at the default dispatch: up to about 8.4 M nonzeros at
 * width 256, 134 M at width 16).  Nothing in the array has to start at zero
 * (every slot the fix-up pass reads is written by the main pass of the same
 * call), so the contents a previous call left behind are harmless; but all calls
 * on one GPU share the one array: calls that overlap on the same GPU (two
 * streams, two threads) race.  Same results, bit for bit, as gcnx_spmm_csr_f32.
 * The array's device address is looked up once per device, by the first
 * gcnx_spmm_chunk_table call on that device (set-up: every matrix needs its
 * chunk table before its first product, and nobody builds it under capture) or,
 * failing that, by the first call of this entry point -- that one look-up loads
 * the code object and may allocate inside the runtime, so a caller that records
 * this entry point in a HIP graph makes one of those two calls on the device
 * before the capture starts.  Every later call reads the published pointer
 * (one atomic load, no lock). */
#define GCNX_SPMM_STATIC_SCRATCH_BYTES (32u << 20)
int gcnx_spmm_csr_static_f32(const gcnx_csr *a, const int32_t *chunk_row,
                             const float *B, int64_t ldb, float *C, int64_t ldc,
                             int32_t width, float alpha, float beta,
                             const uint8_t *val_mask, const int64_t *mask_index,
                             float mask_scale, int fuse_relu, void *stream);

/* The same product with the dense operand in two pieces: rows [0, split_col) of
 * the operand are read from B, rows [split_col, n_cols) from B2 (row 0 of B2 is
 * operand row split_col).  Used by the row-partitioned engine: B holds the
 * local rows of the activation, B2 the halo rows received from the other GPUs,
 * so no combined copy is ever assembled.  No counterpart in the reference
 * (single process). */
int gcnx_spmm_csr_split_f32(const gcnx_csr *a, const int32_t *chunk_row,
                            const float *B, int64_t ldb, const float *B2,
                            int64_t ldb2, int64_t split_col, float *C,
                            int64_t ldc, int32_t width, float alpha, float beta,
                            const uint8_t *val_mask, const int64_t *mask_index,
                            float mask_scale, int fuse_relu,
                            const gcnx_dropout *out_dropout /* nullable */,
                            void *ws, size_t ws_bytes, void *stream);

/* The split product that also writes the sign pattern of what it stores:
 * bit (c % 32) of sign_bits[r * ld_words + c / 32] is set iff stored element
 * (r, c) > 0 (after activation and dropout).  With the relu + dropout epilogue
 * this is the whole gate the backward pass needs (gcnx_gemm_bitgated_f32), 32
 * bytes per 256-wide row instead of re-reading the 1-KiB row.  Widths that are
 * multiples of 32 (gcnx_spmm_signs_supported), 16-byte aligned operands; no value mask.
 * replaces: the relu and dropout gradients of gcn/layers.py:226,210 created at
 * gcn/models.py:58 (their masks). */
int gcnx_spmm_signs_supported(int32_t width);
int gcnx_spmm_csr_signs_f32(const gcnx_csr *a, const int32_t *chunk_row,
                            const float *B, int64_t ldb, const float *B2,
                            int64_t ldb2, int64_t split_col, float *C,
                            int64_t ldc, int32_t width, float alpha, float beta,
                            int fuse_relu, const gcnx_dropout *out_dropout,
                            uint32_t *sign_bits, int64_t ld_words, void *ws,
                            size_t ws_bytes, void *stream);

/* gcnx_spmm_csr_signs_f32 for an output that holds a SUBSET of the rows of the
 * activation the dropout mask is defined on: `a` has one row per output row
 * (e.g. the rows of A_hat that some later product actually reads, stored next to
 * each other), and output row r draws the mask of activation row
 * dropout_row_ids[r] -- element (r, c) is kept iff the Philox uniform of
 * (seed, offset + dropout_row_ids[r] * ld + col0 + c) is < keep -- so the rows
 * come out exactly as the full product would have produced them.
 * dropout_row_ids == NULL: as gcnx_spmm_csr_signs_f32.  sign_bits is nullable
 * here (any width then).  No counterpart in the reference, which always forms
 * every row (gcn/layers.py:203-226). */
int gcnx_spmm_csr_rows_f32(const gcnx_csr *a, const int32_t *chunk_row,
                           const float *B, int64_t ldb, const float *B2,
                           int64_t ldb2, int64_t split_col, float *C,
                           int64_t ldc, int32_t width, float alpha, float beta,
                           int fuse_relu, const gcnx_dropout *out_dropout,
                           const int32_t *dropout_row_ids,
                           uint32_t *sign_bits /* nullable */, int64_t ld_words,
                           void *ws, size_t ws_bytes, void *stream);

/* SpMM with a FACTORED dense operand:  C = A * G  where row c of G is never
 * stored but formed for every nonzero that references it,
 *   G[c, j] = bit(c, j) ? gate_scale * sum_{k < inner} D[c, k] * op(W)[k, j] : 0,
 * bit(c, j) = bit (j % 32) of gate_bits[c * ld_words + j / 32]; op(W) = W
 * ([inner, width], ld ldw) or, with trans_w, W^T (W stored [width, inner]).
 * Rows c >= split_col of D / gate_bits come from D2 / gate_bits2 (row
 * c - split_col): the halo rows received from peers.  This is the backward
 * pass  A_hat^T * ((H' > 0) ? (dP2 * W2^T) / keep : 0)  in one kernel: 96 bytes
 * fetched per nonzero instead of a 1-KiB row of a stored intermediate, and the
 * gated product (gcnx_gemm_bitgated_f32) is not run at all.  The k-sum is the
 * chain of fused multiply-adds of the two-kernel path and rows are summed in the
 * same order; gate_scale multiplies the matrix value rather than the element,
 * so both give the same bits when gate_scale is a power of two and agree to one
 * rounding per term otherwise.  With 16-byte aligned D / gate words (ldd and
 * ld_words multiples of 4) the rows are formed on the matrix pipe, two chunks
 * per wavefront.  Served for inner == 16, width >= 64,
 * width % 32 == 0 (gcnx_spmm_factored_supported); workspace as for
 * gcnx_spmm_csr_f32 at this width.
 * replaces: the gradient of tf.sparse_tensor_dense_matmul composed with the
 * relu / dropout / tf.matmul gradients of layer 2 (gcn/layers.py:210,215,219,226
 * as differentiated at gcn/models.py:58). */
int gcnx_spmm_factored_supported(int32_t width, int32_t inner);
int gcnx_spmm_csr_factored_f32(const gcnx_csr *a, const int32_t *chunk_row,
                               const float *D, int64_t ldd,
                               const uint32_t *gate_bits, int64_t ld_words,
                               const float *D2, int64_t ldd2,
                               const uint32_t *gate_bits2, int64_t ld_words2,
                               int64_t split_col, const float *W, int64_t ldw,
                               int trans_w, int32_t inner, float gate_scale,
                               float *C, int64_t ldc, int32_t width, void *ws,
                               size_t ws_bytes, void *stream);

/* ---- halo exchange of a row-partitioned graph (no counterpart in the
 * reference, which is single-process) ---------------------------------------
 * gcnx_halo_plan: `a` holds local rows [row_begin, row_begin + n_local) with
 * GLOBAL column ids.  Writes (1) need[]: the sorted list of the distinct
 * global columns outside the local range that the shard references (*h_n_need
 * of them; they are the rows to fetch from their owners), and (2)
 * col_remapped[slot]: local columns become c - row_begin, remote ones
 * n_local + position in need[].  GCNX_ERR_WORKSPACE if need_cap is too small
 * (*h_n_need then holds the required size).
 * gcnx_gather_rows_f32: dst[j, 0:width] = src[idx[j], 0:width] (packs the rows
 * a peer asked for into a contiguous send buffer).
 * gcnx_scatter_rows_f32: dst[idx[j], 0:width] = src[j, 0:width] for distinct
 * idx[j] (places received rows into a resident halo buffer whose other rows
 * keep their contents: the labelled rows of the loss gradient, the only ones
 * that are not exactly zero). */
size_t gcnx_halo_plan_workspace_bytes(int64_t n_cols_global);
int gcnx_halo_plan(const gcnx_csr *a, int64_t row_begin, int64_t n_local,
                   int32_t *col_remapped, int32_t *need, int64_t need_cap,
                   int64_t *h_n_need, void *ws, size_t ws_bytes, void *stream);
int gcnx_gather_rows_f32(int64_t n_idx, const int32_t *idx, const float *src,
                         int64_t lds, int32_t width, float *dst, int64_t ldd,
                         void *stream);
/* Cache hint of the wide SpMM (gcnx_csr::gather_keep; no counterpart in the
 * reference).  gcnx_csr_column_degrees: degree[c] = number of nonzeros in column c
 * (degree is zeroed first).  gcnx_csr_gather_keep: keep bit of nonzero k =
 * (degree[col_idx[k]] >= min_degree); bits past nnz in the last word are clear. */
int gcnx_csr_column_degrees(const gcnx_csr *a, int32_t *degree, void *stream);
/* *h_min_degree (host) = the smallest degree d >= 2 such that at most max_columns
 * columns have degree >= d (capped at 4095): the threshold that makes the kept
 * rows fit a cache budget of max_columns rows.  Synchronises the stream. */
size_t gcnx_csr_degree_threshold_workspace_bytes(void);
int gcnx_csr_degree_threshold(const int32_t *degree, int64_t n_cols, int64_t max_columns,
                              int32_t *h_min_degree, void *ws, size_t ws_bytes, void *stream);
int gcnx_csr_gather_keep(const gcnx_csr *a, const int32_t *degree, int32_t min_degree,
                         uint64_t *keep, void *stream);
int gcnx_scatter_rows_f32(int64_t n_idx, const int32_t *idx, const float *src,
                          int64_t lds, int32_t width, float *dst, int64_t ldd,
                          void *stream);
/* Sub-matrix of the nonzeros whose column is flagged (col_keep[j] != 0, one byte
 * per column), same shape, kept nonzeros in their order.  For a product whose
 * dense operand has rows known to be exactly zero: the gradient of the masked
 * loss (gcn/metrics.py:4-11, mask / mean(mask) multiplies the per-row loss) is
 * zero outside the labelled rows, so A_hat^T dZ only meets the labelled columns
 * (the reference multiplies the zeros, tf.gradients of gcn/layers.py:222-226).
 * Two calls around the caller's allocation: _count fills out_row_ptr
 * [n_rows + 1] and *h_nnz (host; synchronises the stream), _fill writes the
 * out_nnz = *h_nnz kept entries (out_src[k], optional: the slot of entry k in a)
 * from the workspace _count left behind, which must not be touched in between. */
size_t gcnx_csr_filter_workspace_bytes(int64_t nnz);
int gcnx_csr_filter_columns_count(const gcnx_csr *a, const uint8_t *col_keep, int64_t *out_row_ptr,
                                  int64_t *h_nnz, void *ws, size_t ws_bytes, void *stream);
int gcnx_csr_filter_columns_fill(const gcnx_csr *a, const void *ws, size_t ws_bytes, int64_t out_nnz,
                                 int32_t *out_col, float *out_val, int64_t *out_src, void *stream);

/* ---- the exchange itself, over RCCL (xGMI inside a node) ------------------
 * No counterpart in the reference (single process).  One process per GPU; every
 * process holds one communicator for its device.  RCCL is bound at run time
 * (dlopen), so the library also loads where RCCL is absent:
 * gcnx_comm_available() says whether these entry points can work here.
 * `comm` is an ncclComm_t passed as void*: either one the host already owns
 * (ncclCommInitRank) or one made by gcnx_comm_create from the 128-byte id that
 * rank 0 draws with gcnx_comm_unique_id and hands to every rank by any means
 * (the Python host broadcasts it through torch.distributed).  gcnx_comm_create
 * is collective over the ranks and binds the communicator to the CURRENT device.
 *
 * gcnx_halo_allgather_f32: every rank receives the rows of the dense operand
 * that its CSR block references on other ranks (need[] of gcnx_halo_plan) and
 * sends the rows the others asked for.  Two steps on `stream`: (1) with
 * plan->send_idx, pack send_buf[j, 0:ncols] = local[send_idx[j], col0:col0+ncols]
 * (send_idx NULL: send_buf already holds the packed rows); (2) one ncclSend /
 * ncclRecv per peer inside one group, so that all pairs of GPUs exchange
 * concurrently, each over its own link.  halo receives h_recv_rows[p] rows of
 * ncols floats from peer p = 0, 1, ..., world-1 in that order, which is the
 * order of need[] (sorted by global row id, hence grouped by owner).  Rows are
 * moved as bytes: 32-bit sign words travel through the same call.  Asynchronous
 * and stream-ordered like the compute entry points: issue it on a stream of its
 * own to overlap it with compute and order the two with events.  Calls on one
 * communicator must be issued in the same order on every rank.
 * gcnx_comm_allreduce_sum_f32: in-place sum over the ranks (weight gradients,
 * loss). */
#define GCNX_COMM_ID_BYTES 128
typedef struct gcnx_halo_exchange {
  int32_t world;              /* ranks of the communicator                     */
  int32_t rank;               /* this rank                                     */
  const int64_t *h_send_rows; /* HOST [world]: rows this rank sends to peer p  */
  const int64_t *h_recv_rows; /* HOST [world]: rows it receives from peer p    */
  const int32_t *send_idx;    /* DEVICE [sum h_send_rows]: local row of every
                                 row to send, grouped by destination; nullable */
} gcnx_halo_exchange;
int gcnx_comm_available(void);
int gcnx_comm_unique_id(void *id_out /* GCNX_COMM_ID_BYTES, host */);
int gcnx_comm_create(int32_t world, int32_t rank, const void *id, void **comm_out);
int gcnx_comm_destroy(void *comm);
int gcnx_halo_allgather_f32(const gcnx_halo_exchange *plan, void *comm,
                            const float *local, int64_t ld_local, int64_t col0,
                            int32_t ncols, float *send_buf, float *halo,
                            void *stream);
int gcnx_comm_allreduce_sum_f32(void *comm, float *buf, int64_t n, void *stream);

/* Measurement hook (no counterpart in the reference): while set, every SpMM
 * call made by this thread records hipEvent `start` right before its first
 * kernel and `stop` right after its last one (the main pass and, when rows are
 * cut by chunk boundaries, the fix-up pass), on the launch stream, so a
 * benchmark can time the product alone inside a longer timed region.  The
 * setting is thread-local.  Pass NULL, NULL to clear. */
void gcnx_profile_spmm_events(void *start, void *stop);

/* ---- Multi-support sum  C = act(sum_i A_i * P_i)  as ONE product -----------
 * replaces: the loop over supports plus tf.add_n, gcn/layers.py:213-221 (and
 * :195-198).  The supports are merged into one COO over interleaved columns,
 * col' = col * n_sup + i, so that  sum_i A_i * P_i = A_cat * reshape(P_cat)
 * where P_cat[N, n_sup*width] = x * [W_0 | ... | W_{n-1}]; the result is a
 * plain gcnx_spmm_csr_f32 call that accumulates across supports in registers.
 * This entry point writes the merged coordinates; h_* arrays are HOST arrays of
 * n_sup device pointers / sizes.  Outputs have sum(h_nnz) entries.
 */
int gcnx_coo_interleave_supports(int32_t n_sup, const int64_t *h_nnz,
                                 const int32_t *const *h_row,
                                 const int32_t *const *h_col,
                                 int64_t coord_stride,
                                 const void *const *h_val, int val_is_f64,
                                 int32_t *out_row, int32_t *out_col,
                                 void *out_val, void *stream);

/* ---- dense product  C = alpha * op(A) * op(B) + beta * C -------------------
 * replaces: dot(x, y, sparse=False) = tf.matmul, gcn/layers.py:35 (call sites
 * :215 for H' * W and :123), and its two gradients H'^T * dP and dP * W^T
 * (gcn/models.py:58).  fp32 in, fp32 accumulate.  op(A) is M x K, op(B) K x N.
 * The K-reduction of a transposed-A product over a very tall A is split across
 * workgroups and completed from partial tiles in a fixed order (ws).
 */
size_t gcnx_gemm_workspace_bytes(int trans_a, int trans_b, int64_t M, int64_t N,
                                 int64_t K);
int gcnx_gemm_f32(int trans_a, int trans_b, int64_t M, int64_t N, int64_t K,
                  const float *A, int64_t lda, const float *B, int64_t ldb,
                  float *C, int64_t ldc, float alpha, float beta, int fuse_relu,
                  void *ws, size_t ws_bytes, void *stream);

/* Arithmetic of the large dense products (thread-local setting, like the
 * profiling hook; returns the previous mode; any other argument only reads).
 * 0 (default): products that fill the matrix pipe run on the bf16 MFMA with every
 * fp32 operand split exactly into three bf16 pieces and the six partial products
 * of weight >= 2^-16 accumulated in fp32; the dropped ones are below 2^-23 of each
 * product, the size of one fp32 rounding, so results carry fp32 accuracy (same
 * 1e-5 bar against the fp64 yardstick as every other kernel) but are not the bits
 * of an fp32 multiply-add chain.  1: fp32-input MFMA chains (v_mfma_f32_32x32x2_f32)
 * only -- exact fp32 arithmetic, at the fp32 vector rate.  Small, skinny and
 * K <= 32 products always use exact fp32 arithmetic. */
int gcnx_dense_arith(int mode);

/* The same product with dropout applied to A while its tiles are staged, so
 * dropout(A) is never written: replaces tf.nn.dropout(x) followed by
 * tf.matmul(x', W) (gcn/layers.py:210,215) and, with trans_a, the weight
 * gradient x'^T * dP (gcn/models.py:58) on the same regenerated mask.  The
 * descriptor indexes A AS STORED: element (r, c) has Philox index
 * offset + r * ld + c (col0 must be 0).  Only the MFMA kernels implement it
 * (tall A, one weight dimension >= 64); otherwise GCNX_ERR_UNSUPPORTED. */
int gcnx_gemm_dropout_a_supported(int trans_a, int trans_b, int64_t M, int64_t N,
                                  int64_t K); /* 1 if the shape is served */
int gcnx_gemm_dropout_a_f32(int trans_a, int trans_b, int64_t M, int64_t N, int64_t K,
                            const float *A, int64_t lda,
                            const gcnx_dropout *a_dropout, const float *B,
                            int64_t ldb, float *C, int64_t ldc, float alpha,
                            float beta, void *ws, size_t ws_bytes, void *stream);

/* The same product with a gate in the epilogue: C = gate > 0 ? C * gate_scale : 0
 * element-wise, gate an M x N matrix.  With gate = dropout(relu(.)) of the
 * forward pass and gate_scale = 1/keep this is the backward of relu followed by
 * dropout (gcn/layers.py:226,210) applied to dP * W^T without a separate pass:
 * dropout(relu(h)) > 0 exactly where h > 0 and the element was kept. */
int gcnx_gemm_gated_f32(int trans_a, int trans_b, int64_t M, int64_t N, int64_t K,
                        const float *A, int64_t lda, const float *B, int64_t ldb,
                        float *C, int64_t ldc, float alpha, float beta,
                        const float *gate, int64_t ldgate, float gate_scale,
                        void *ws, size_t ws_bytes, void *stream);

/* The product followed by relu (fuse_relu != 0) and dropout on the output:
 * element (m, n) is kept iff the Philox uniform at index out_dropout->offset +
 * m * out_dropout->ld + out_dropout->col0 + n is < keep (the mask of
 * gcnx_bernoulli_mask_u8 and of the SpMM epilogue), survivors scaled by 1/keep.
 * replaces: act(.) of one layer followed by tf.nn.dropout at the head of the
 * next (gcn/layers.py:226 then :210) when the layer ends in the dense product,
 * i.e. when a caller evaluates (A_hat x) W instead of A_hat (x W).  Served by
 * the tall MFMA kernel only (gcnx_gemm_relu_dropout_supported; rows of C and the
 * descriptor's offset / ld / col0 multiples of 4, beta = 0); other shapes return
 * GCNX_ERR_UNSUPPORTED and the caller applies gcnx_dropout_philox_f32 afterwards. */
int gcnx_gemm_relu_dropout_supported(int trans_a, int trans_b, int64_t M, int64_t N, int64_t K);
int gcnx_gemm_relu_dropout_f32(int trans_a, int trans_b, int64_t M, int64_t N, int64_t K,
                               const float *A, int64_t lda, const float *B, int64_t ldb,
                               float *C, int64_t ldc, float alpha, float beta,
                               int fuse_relu, const gcnx_dropout *out_dropout,
                               void *ws, size_t ws_bytes, void *stream);

/* gcnx_gemm_relu_dropout_f32 that also writes the sign pattern of the stored C
 * (bit (n % 32) of sign_bits[m * ld_words + n / 32] set <=> element (m, n) > 0),
 * the gate the backward pass needs (gcnx_gemm_bitgated_f32): the aggregate-first
 * counterpart of gcnx_spmm_csr_signs_f32.  Needs keep < 1 and N % 32 == 0 on top
 * of the conditions of gcnx_gemm_relu_dropout_f32.
 * replaces: the relu and dropout gradients of gcn/layers.py:226,210 (their masks). */
int gcnx_gemm_relu_dropout_signs_f32(int trans_a, int trans_b, int64_t M, int64_t N, int64_t K,
                                     const float *A, int64_t lda, const float *B, int64_t ldb,
                                     float *C, int64_t ldc, float alpha, float beta,
                                     int fuse_relu, const gcnx_dropout *out_dropout,
                                     uint32_t *sign_bits, int64_t ld_words,
                                     void *ws, size_t ws_bytes, void *stream);

/* The gate as one bit per element: bit (n % 32) of gate_bits[m * ld_words +
 * n / 32] set <=> keep.  Used by the row-partitioned engine for the halo rows
 * of d(A P1): only the sign pattern of the remote rows of H' travels (32 bytes
 * per 256-wide row instead of 1 KiB), the gated product is recomputed locally
 * from the remote rows of dP2.  No counterpart in the reference (single process). */
int gcnx_gemm_bitgated_f32(int trans_a, int trans_b, int64_t M, int64_t N, int64_t K,
                           const float *A, int64_t lda, const float *B, int64_t ldb,
                           float *C, int64_t ldc, float alpha, float beta,
                           const uint32_t *gate_bits, int64_t ld_words,
                           float gate_scale, void *ws, size_t ws_bytes, void *stream);

/* bits[i * ld_words + w] = sign pattern (element > 0) of x[row(i)][32w .. 32w+31],
 * row(i) = row_idx ? row_idx[i] : i; columns beyond width read as 0. */
int gcnx_pack_sign_bits_f32(int64_t n_rows, const int32_t *row_idx, const float *x,
                            int64_t ldx, int32_t width, uint32_t *bits,
                            int64_t ld_words, void *stream);

/* gcnx_gemm_dropout_a_f32 for a plain (not transposed) A that holds a gathered
 * subset of the rows the mask is defined on: row i of A draws the mask of row
 * a_row_ids[i], i.e. element (i, c) has index a_dropout->offset + a_row_ids[i] *
 * a_dropout->ld + c.  Lets a GPU recompute dropout(X) W for the remote rows of
 * X it holds a copy of (the features are constant) instead of receiving them. */
int gcnx_gemm_dropout_a_rows_f32(int trans_b, int64_t M, int64_t N, int64_t K,
                                 const float *A, int64_t lda,
                                 const gcnx_dropout *a_dropout,
                                 const int32_t *a_row_ids, const float *B,
                                 int64_t ldb, float *C, int64_t ldc, float alpha,
                                 float beta, void *ws, size_t ws_bytes, void *stream);

/* The same for the transposed form C[M,N] = dropout(A)^T B with A stored [K, M]:
 * row k of A draws the mask of row a_row_ids[k] (K ids).  For the weight gradient
 * X'^T dP1 summed over the rows of dP1 that can be nonzero only: X holds those
 * rows of the features, gathered once.  Served where
 * gcnx_gemm_dropout_at_rows_supported says so (split-operand arithmetic, N a
 * multiple of 256, the tall transposed MFMA shape); GCNX_ERR_UNSUPPORTED otherwise. */
int gcnx_gemm_dropout_at_rows_supported(int64_t M, int64_t N, int64_t K);
int gcnx_gemm_dropout_at_rows_f32(int64_t M, int64_t N, int64_t K, const float *A, int64_t lda,
                                  const gcnx_dropout *a_dropout, const int32_t *a_row_ids,
                                  const float *B, int64_t ldb, float *C, int64_t ldc, float alpha,
                                  float beta, void *ws, size_t ws_bytes, void *stream);

/* ---- glue kernels of one training step ------------------------------------ */

/* replaces: masked_softmax_cross_entropy + masked_accuracy, gcn/metrics.py:4-20.
 * out_loss_acc[0] = mean_i(CE_i * mask_i / mean(mask)), out_loss_acc[1] = masked
 * accuracy (first maximum wins), grad (nullable) = d loss / d logits.
 * mask_total <= 0: the masked rows are counted here.  mask_total > 0: it is the
 * count over ALL row shards of a row-partitioned run; the outputs are then this
 * shard's share (sum over shards = the loss / accuracy) and grad is final.
 * ws: gcnx_xent_workspace_bytes(n). */
size_t gcnx_xent_workspace_bytes(int64_t n);
int gcnx_masked_softmax_xent_f32(int64_t n, int32_t n_class, const float *logits,
                                 int64_t ld, const float *labels, int64_t ldl,
                                 const uint8_t *mask, double mask_total,
                                 float *out_loss_acc, float *grad, int64_t ldg,
                                 void *ws, size_t ws_bytes, void *stream);

/* replaces: tf.nn.dropout on a dense activation, gcn/layers.py:210,120.
 * out = mask ? x * scale : 0; mask is one byte per element (the caller draws
 * it, so a parity run can share it).  In place allowed. */
int gcnx_dropout_dense_f32(int64_t n, const float *x, const uint8_t *mask,
                           float scale, float *out, void *stream);

/* Counter-based Bernoulli(keep) mask: byte k = (uniform(seed, offset + k) <
 * keep).  Philox4x32-10; the same (seed, offset) gives the same mask on any
 * device count.  replaces: tf.random_uniform in gcn/layers.py:24. */
int gcnx_bernoulli_mask_u8(int64_t n, float keep, uint64_t seed,
                           uint64_t offset, uint8_t *mask, void *stream);

/* Same mask with the offset advanced on the device: effective offset = offset
 * + *step_dev * step_stride, read when the kernel runs.  A launch recorded in a
 * HIP graph therefore draws a fresh mask on every replay, as session.run does
 * on every call of gcn/train_normal.py:93 (each run re-samples dropout). */
int gcnx_bernoulli_mask_dev_u8(int64_t n, float keep, uint64_t seed,
                               uint64_t offset, const uint64_t *step_dev,
                               uint64_t step_stride, uint8_t *mask, void *stream);

/* *counter += inc on the stream (one lane).  The step counters read by
 * gcnx_bernoulli_mask_dev_u8 and gcnx_adam_step_dev_f32 are advanced with it so
 * that the increment is part of a captured graph. */
int gcnx_counter_add_u64(uint64_t *counter, uint64_t inc, void *stream);

/* Synthetic dense inputs (no counterpart in the reference): out[i] = lo +
 * (hi - lo) * u(seed, offset + i) with the same Philox uniform as the masks, a
 * pure function of the global element index, so that features and labels are
 * the same for every GPU count. */
int gcnx_philox_uniform_f32(int64_t n, float lo, float hi, uint64_t seed,
                            uint64_t offset, float *out, void *stream);

/* The same two steps fused, the mask never touching memory: element i is kept
 * iff the Philox uniform of (seed, offset + i) is < keep (bit-identical to
 * gcnx_bernoulli_mask_u8), survivors scaled by 1/keep.  mask_out (nullable)
 * also stores the mask.  replaces: gcn/layers.py:210,120 + :24. */
int gcnx_dropout_philox_f32(int64_t n, const float *x, float keep, uint64_t seed,
                            uint64_t offset, float *out, uint8_t *mask_out,
                            void *stream);

/* The same dropout for a gathered subset of rows of a matrix with rows of
 * `width` elements: out[i][c] = keep(offset + row_ids[i] * width + c) ?
 * x[i][c] / keep : 0 (row i of x draws the mask of row row_ids[i]).  out may be
 * x (in place).  Used for the resident copies of remote feature rows in the
 * row-partitioned engine and for activations kept over a subset of the rows. */
int gcnx_dropout_rows_philox_f32(int64_t n_rows, int32_t width, const float *x,
                                 int64_t ldx, const int32_t *row_ids, float keep,
                                 uint64_t seed, uint64_t offset, float *out,
                                 int64_t ldo, void *stream);

/* Backward of relu followed by that dropout, mask regenerated from
 * (seed, offset): dx = (y > 0 && kept_i) ? dy / keep : 0; y = relu output. */
int gcnx_relu_dropout_bwd_philox_f32(int64_t n, const float *y, const float *dy,
                                     float keep, uint64_t seed, uint64_t offset,
                                     float *dx, void *stream);

/* dx = (y > 0 ? dy : 0) [* mask * scale]: backward of act=relu
 * (gcn/layers.py:226) fused with the backward of the next layer's dropout
 * (gcn/layers.py:210).  y is the relu OUTPUT.  mask nullable. */
int gcnx_relu_dropout_bwd_f32(int64_t n, const float *y, const float *dy,
                              const uint8_t *mask, float scale, float *dx,
                              void *stream);

/* replaces: tf.train.AdamOptimizer(lr).minimize, gcn/models.py:142,58, with
 * weight decay folded in as grad + weight_decay * w (gradient of
 * weight_decay * l2_loss(w), gcn/models.py:148-149).  t counts from 1. */
int gcnx_adam_step_f32(int64_t n, float *w, const float *grad, float *m,
                       float *v, float lr, float beta1, float beta2, float eps,
                       float weight_decay, int64_t t, void *stream);

/* replaces: tf.nn.l2_loss(var) and its addition into the loss,
 * gcn/models.py:148-149.  out[0] = (add ? add[0] : 0) + scale * sum(w^2) / 2;
 * add (nullable, one device float) lets the caller chain the weight-decay terms
 * onto the cross-entropy without separate scalar kernels; add == out is
 * allowed.  One fixed reduction tree: reproducible.  The gradient of this term
 * is not produced here: gcnx_adam_step_f32 folds it in (weight_decay). */
int gcnx_l2_loss_f32(int64_t n, const float *w, float scale, const float *add,
                     float *out, void *stream);

/* Same update with the step count t (>= 1) read from device memory when the
 * kernel runs (the bias correction is formed on the device), for launches
 * recorded in a HIP graph.  replaces: gcn/models.py:142,58. */
int gcnx_adam_step_dev_f32(int64_t n, float *w, const float *grad, float *m,
                           float *v, float lr, float beta1, float beta2,
                           float eps, float weight_decay, const uint64_t *t_dev,
                           void *stream);

/* ---- degree normalisation on the device -----------------------------------
 * replaces: normalize_adj / preprocess_adj, gcn/utils.py:122-135, for graphs
 * too large for the host.  `a` holds local rows [row_begin, row_begin+n_rows)
 * of a matrix with a symmetric pattern and symmetric values (A + I of an
 * undirected graph).  The entry at (r, c) becomes
 *   fp32( fp64( fp64(a * d[r]) * d[c] ) ),  d[i] = d_table[row_sum[i]],
 * the two fp64 roundings of the reference's (A.D)^T.D.  d_table is computed on
 * the host as numpy.power(k, -0.5) for the integer row sums k (entry 0 = 0, the
 * reference's inf -> 0 rule); row_sum is indexed by GLOBAL vertex id.
 * *status_flag (device int, zeroed by the caller) is set to 1 if a row sum
 * falls outside the table.
 */
int gcnx_normalize_csr_f64(const gcnx_csr *a, int64_t row_begin,
                           const int32_t *row_sum, const double *d_table,
                           int64_t d_table_len, float *val_out,
                           int *status_flag, void *stream);

/* ---- synthetic RMAT graphs --------------------------------------------------
 * No counterpart in the reference (its inputs are the Planetoid files); this is
 * the generator of the benchmark inputs.  Edge e of 2^scale * edge_factor
 * undirected edges is drawn from Philox4x32-10 keyed by (seed, e) with
 * quadrant probabilities (a, b, c, 1-a-b-c), vertex ids are scrambled by a
 * seeded bijection, self-loops dropped.  Writes, for the rows
 * [row_begin, row_end) only, both directions of every edge as 64-bit keys
 * (row << 32 | col), unsorted and with duplicates, into keys_out (capacity
 * `capacity`); *h_count receives the number written (or needed, if larger than
 * capacity, in which case the status is GCNX_ERR_WORKSPACE).
 */
int gcnx_rmat_edges(int32_t scale, int64_t edge_begin, int64_t edge_end,
                    double a, double b, double c, uint64_t seed,
                    int64_t row_begin, int64_t row_end, uint64_t *keys_out,
                    int64_t capacity, int64_t *h_count, void *ws, void *stream);
size_t gcnx_rmat_workspace_bytes(void);

/* Sort + de-duplicate 64-bit (row << 32 | col) keys, add the diagonal (A + I),
 * and emit a CSR over local rows [row_begin, row_end) with all values 1.0.
 * *h_nnz receives the number of entries.  col_idx/val capacity = n_keys +
 * (row_end - row_begin). */
size_t gcnx_csr_from_keys_workspace_bytes(int64_t n_keys, int64_t n_local_rows);
int gcnx_csr_from_keys(uint64_t *keys, int64_t n_keys, int64_t row_begin,
                       int64_t row_end, int add_identity, int64_t *row_ptr,
                       int32_t *col_idx, float *val, int64_t *h_nnz, void *ws,
                       size_t ws_bytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* GCNX_H */
