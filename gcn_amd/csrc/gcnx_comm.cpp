// Halo exchange and gradient all-reduce of the row-partitioned engine over RCCL
// (xGMI inside a node), behind the C ABI so that a host without Python can drive
// the multi-GPU path.
//
// The exchange is a variable-size all-to-all written as grouped point-to-point
// transfers: inside one ncclGroupStart / ncclGroupEnd every rank posts one
// ncclSend and one ncclRecv per peer it trades rows with, so all pairs move
// their piece concurrently, each pair over its own xGMI link.  (A ring
// collective would push every byte through one link per hop.)
//
// RCCL is bound at run time (dlopen / dlsym): the library has no link-time
// dependency on it, loads on a machine without RCCL (every entry point here then
// returns GCNX_ERR_UNSUPPORTED), and a process that already holds a copy of
// librccl.so -- PyTorch ships one -- shares that copy instead of loading a
// second one.
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <string.h>

#include <mutex>

#include "gcnx_common.h"

namespace {

struct Rccl {
  void *handle = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclCommCount) CommCount = nullptr;
  decltype(&ncclCommUserRank) CommUserRank = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclSend) Send = nullptr;
  decltype(&ncclRecv) Recv = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  bool ok = false;
  char why[256] = "";   // loader message when ok is false
};

std::once_flag g_once;
Rccl g_rccl;

template <typename F>
bool bind(void *h, const char *name, F *out) {
  *out = reinterpret_cast<F>(dlsym(h, name));
  return *out != nullptr;
}

void load_rccl() {
  // a copy this process already holds first (same name torch's libraries ask for)
  const char *names[] = {"librccl.so", "librccl.so.1"};
  void *h = nullptr;
  for (const char *n : names)
    if (!h) h = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
  for (const char *n : names)
    if (!h) h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
  Rccl &r = g_rccl;
  if (!h) {
    const char *msg = dlerror();
    snprintf(r.why, sizeof(r.why), "%s", msg ? msg : "dlopen failed");
    return;
  }
  r.handle = h;
  r.ok = bind(h, "ncclGetUniqueId", &r.GetUniqueId) && bind(h, "ncclCommInitRank", &r.CommInitRank) &&
         bind(h, "ncclCommDestroy", &r.CommDestroy) && bind(h, "ncclCommCount", &r.CommCount) &&
         bind(h, "ncclCommUserRank", &r.CommUserRank) && bind(h, "ncclGroupStart", &r.GroupStart) &&
         bind(h, "ncclGroupEnd", &r.GroupEnd) && bind(h, "ncclSend", &r.Send) && bind(h, "ncclRecv", &r.Recv) &&
         bind(h, "ncclAllReduce", &r.AllReduce) && bind(h, "ncclGetErrorString", &r.GetErrorString);
  if (!r.ok) snprintf(r.why, sizeof(r.why), "a required symbol is missing from librccl.so");
}

const Rccl *rccl() {
  std::call_once(g_once, load_rccl);
  return g_rccl.ok ? &g_rccl : nullptr;
}

}  // namespace

#define GCNX_NEED_RCCL(r)                                                                  \
  const Rccl *r = rccl();                                                                  \
  GCNX_REQUIRE(r != nullptr, GCNX_ERR_UNSUPPORTED, "RCCL is not available: %s", g_rccl.why)

#define GCNX_CHECK_RCCL(r, expr)                                                                     \
  do {                                                                                               \
    ncclResult_t e_ = (expr);                                                                        \
    if (e_ != ncclSuccess) {                                                                         \
      gcnx::set_error("%s failed: %s (%s:%d)", #expr, (r)->GetErrorString(e_), __FILE__, __LINE__);  \
      return GCNX_ERR_COMM;                                                                          \
    }                                                                                                \
  } while (0)

static_assert(GCNX_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "unique id size");

extern "C" int gcnx_comm_available(void) { return rccl() ? 1 : 0; }

extern "C" int gcnx_comm_unique_id(void *id_out) {
  GCNX_REQUIRE(id_out, GCNX_ERR_INVALID_ARG, "comm_unique_id: null output");
  GCNX_NEED_RCCL(r);
  ncclUniqueId id;
  GCNX_CHECK_RCCL(r, r->GetUniqueId(&id));
  memcpy(id_out, &id, sizeof(id));
  return GCNX_OK;
}

extern "C" int gcnx_comm_create(int32_t world, int32_t rank, const void *id, void **comm_out) {
  GCNX_REQUIRE(id && comm_out && world >= 1 && rank >= 0 && rank < world, GCNX_ERR_INVALID_ARG,
               "comm_create: bad argument (world %d, rank %d)", world, rank);
  GCNX_NEED_RCCL(r);
  ncclUniqueId uid;
  memcpy(&uid, id, sizeof(uid));
  ncclComm_t c = nullptr;
  GCNX_CHECK_RCCL(r, r->CommInitRank(&c, world, uid, rank));
  *comm_out = c;
  return GCNX_OK;
}

extern "C" int gcnx_comm_destroy(void *comm) {
  if (!comm) return GCNX_OK;
  GCNX_NEED_RCCL(r);
  GCNX_CHECK_RCCL(r, r->CommDestroy(static_cast<ncclComm_t>(comm)));
  return GCNX_OK;
}

extern "C" int gcnx_halo_allgather_f32(const gcnx_halo_exchange *plan, void *comm, const float *local,
                                       int64_t ld_local, int64_t col0, int32_t ncols, float *send_buf,
                                       float *halo, void *stream) {
  GCNX_REQUIRE(plan && comm, GCNX_ERR_INVALID_ARG, "halo_allgather: null plan or communicator");
  GCNX_REQUIRE(plan->world >= 1 && plan->rank >= 0 && plan->rank < plan->world && plan->h_send_rows &&
                   plan->h_recv_rows && ncols >= 0 && col0 >= 0,
               GCNX_ERR_INVALID_ARG, "halo_allgather: bad plan");
  GCNX_NEED_RCCL(r);
  ncclComm_t c = static_cast<ncclComm_t>(comm);
  int n_ranks = 0, my_rank = -1;
  GCNX_CHECK_RCCL(r, r->CommCount(c, &n_ranks));
  GCNX_CHECK_RCCL(r, r->CommUserRank(c, &my_rank));
  GCNX_REQUIRE(n_ranks == plan->world && my_rank == plan->rank, GCNX_ERR_INVALID_ARG,
               "halo_allgather: plan is for rank %d of %d, communicator is rank %d of %d", plan->rank, plan->world,
               my_rank, n_ranks);
  int64_t n_send = 0, n_recv = 0;
  for (int p = 0; p < plan->world; ++p) {
    GCNX_REQUIRE(plan->h_send_rows[p] >= 0 && plan->h_recv_rows[p] >= 0, GCNX_ERR_INVALID_ARG,
                 "halo_allgather: negative row count for peer %d", p);
    n_send += plan->h_send_rows[p];
    n_recv += plan->h_recv_rows[p];
  }
  if (ncols == 0 || (n_send == 0 && n_recv == 0)) return GCNX_OK;
  GCNX_REQUIRE((n_send == 0 || send_buf) && (n_recv == 0 || halo), GCNX_ERR_INVALID_ARG,
               "halo_allgather: null buffer");
  hipStream_t st = gcnx::as_stream(stream);
  if (plan->send_idx && n_send) {
    // pack the rows the peers asked for: send_buf[j, :] = local[send_idx[j], col0 : col0 + ncols]
    GCNX_REQUIRE(local && ld_local >= col0 + ncols, GCNX_ERR_INVALID_ARG, "halo_allgather: bad local operand");
    const int rc = gcnx_gather_rows_f32(n_send, plan->send_idx, local + col0, ld_local, ncols, send_buf, ncols, stream);
    if (rc != GCNX_OK) return rc;
  }
  // one send and one receive per trading peer, all in flight together; rows are
  // moved as bytes (sign words travel through the same call)
  const size_t row_bytes = (size_t)ncols * sizeof(float);
  GCNX_CHECK_RCCL(r, r->GroupStart());
  ncclResult_t first_bad = ncclSuccess;
  int64_t send_row = 0, recv_row = 0;
  for (int p = 0; p < plan->world; ++p) {
    const int64_t ns = plan->h_send_rows[p], nr = plan->h_recv_rows[p];
    if (ns > 0) {
      const ncclResult_t e = r->Send(reinterpret_cast<const char *>(send_buf) + (size_t)send_row * row_bytes,
                                     (size_t)ns * row_bytes, ncclInt8, p, c, st);
      if (e != ncclSuccess && first_bad == ncclSuccess) first_bad = e;
    }
    if (nr > 0) {
      const ncclResult_t e = r->Recv(reinterpret_cast<char *>(halo) + (size_t)recv_row * row_bytes,
                                     (size_t)nr * row_bytes, ncclInt8, p, c, st);
      if (e != ncclSuccess && first_bad == ncclSuccess) first_bad = e;
    }
    send_row += ns;
    recv_row += nr;
  }
  // the group is always closed, also after a failed post
  const ncclResult_t end = r->GroupEnd();
  GCNX_CHECK_RCCL(r, first_bad);
  GCNX_CHECK_RCCL(r, end);
  return GCNX_OK;
}

extern "C" int gcnx_comm_allreduce_sum_f32(void *comm, float *buf, int64_t n, void *stream) {
  GCNX_REQUIRE(comm && n >= 0 && (n == 0 || buf), GCNX_ERR_INVALID_ARG, "comm_allreduce: bad argument");
  if (n == 0) return GCNX_OK;
  GCNX_NEED_RCCL(r);
  GCNX_CHECK_RCCL(r, r->AllReduce(buf, buf, (size_t)n, ncclFloat, ncclSum, static_cast<ncclComm_t>(comm),
                                  gcnx::as_stream(stream)));
  return GCNX_OK;
}
