// nif_comm.hip -- the multi-GPU side of the C-ABI (include/nif_hip.h, "multi-GPU" section): RCCL over xGMI, called
// directly (no tensor framework, no MPI).  Replaces the reference's `tf.distribute.MirroredStrategy().scope()`
// recipe (reference README.md:39-49): the point batch is sharded over the GPUs of one node, every rank computes
// loss and gradient of its shard pre-scaled by 1/B_global (nif_loss_grad_dev), and ONE ncclAllReduce(sum, f32,
// P+1) of the flat buffer [grad | loss] per step makes every rank hold the global-batch gradient; every rank then
// applies the identical Adam update (replicated state).  The collective is enqueued on the context's own HIP
// stream: no host synchronisation, no copy.
//
// Two ways to build the communicator, both one nif_ctx per GPU:
//   * one process per GPU (bench.py, Model.fit):  rank 0 calls nif_comm_unique_id, hands the 128 bytes to the
//     other processes (host side: nif_amd/distributed.py), every rank calls nif_comm_init_rank;
//   * one process driving n GPUs (nif_train_step_multi): nif_comm_init_all = ncclCommInitAll.
#include "nif_ctx.h"
#include <rccl/rccl.h>

#include <cstring>
#include <string>
#include <thread>
#include <vector>

#define NCCLCHK(expr)                                                                             \
  do {                                                                                            \
    ncclResult_t r_ = (expr);                                                                     \
    if (r_ != ncclSuccess)                                                                        \
      return fail(NIF_ERR_COMM, std::string(#expr) + ": " + ncclGetErrorString(r_));               \
  } while (0)

static_assert(NIF_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "nif_hip.h: NIF_COMM_ID_BYTES must equal RCCL's unique-id size");

static inline ncclComm_t comm_of(const nif_ctx* c) { return (ncclComm_t)c->comm; }

extern "C" int nif_comm_unique_id(void* id_out) {
  if (!id_out) return fail(NIF_ERR_INVALID, "null");
  ncclUniqueId id;
  NCCLCHK(ncclGetUniqueId(&id));
  memcpy(id_out, &id, NIF_COMM_ID_BYTES);
  return NIF_OK;
}

static int comm_scratch(nif_ctx* c) {
  if (!c->comm_scratch) {
    HIPCHK(hipMalloc(&c->comm_scratch, 64));
    HIPCHK(hipMemsetAsync(c->comm_scratch, 0, 64, c->st));
  }
  return NIF_OK;
}

extern "C" int nif_comm_init_rank(nif_ctx* c, const void* id, int32_t rank, int32_t world) {
  if (!c || !id || world < 1 || rank < 0 || rank >= world) return fail(NIF_ERR_INVALID, "bad argument");
  if (c->comm) return fail(NIF_ERR_STATE, "context already has a communicator");
  HIPCHK(hipSetDevice(c->dev));
  { const int rct = nif_tail_flush(c); if (rct) return rct; }      // (a deferred row reduction: the all-reduce reads [grad | loss])
  ncclUniqueId uid;
  memcpy(&uid, id, NIF_COMM_ID_BYTES);
  ncclComm_t comm = nullptr;
  NCCLCHK(ncclCommInitRank(&comm, world, uid, rank));
  c->comm = comm; c->comm_rank = rank; c->comm_world = world;
  return comm_scratch(c);
}

extern "C" int nif_comm_init_all(nif_ctx** ctxs, int32_t n) {
  if (!ctxs || n < 1 || n > 64) return fail(NIF_ERR_INVALID, "bad argument");
  int devs[64];
  for (int i = 0; i < n; ++i) {
    if (!ctxs[i]) return fail(NIF_ERR_INVALID, "null context");
    if (ctxs[i]->comm) return fail(NIF_ERR_STATE, "context already has a communicator");
    { const int rct = nif_tail_flush(ctxs[i]); if (rct) return rct; }
    devs[i] = ctxs[i]->dev;
    for (int j = 0; j < i; ++j)
      if (devs[j] == devs[i]) return fail(NIF_ERR_INVALID, "two contexts of one communicator on the same device");
  }
  ncclComm_t comms[64];
  NCCLCHK(ncclCommInitAll(comms, n, devs));
  for (int i = 0; i < n; ++i) {
    ctxs[i]->comm = comms[i]; ctxs[i]->comm_rank = i; ctxs[i]->comm_world = n;
    HIPCHK(hipSetDevice(ctxs[i]->dev));
    int rc = comm_scratch(ctxs[i]); if (rc) return rc;
  }
  return NIF_OK;
}

extern "C" int nif_comm_destroy(nif_ctx* c) {
  if (!c) return fail(NIF_ERR_INVALID, "null");
  if (!c->comm) return NIF_OK;
  HIPCHK(hipSetDevice(c->dev));
  HIPCHK(hipStreamSynchronize(c->st));
  ncclComm_t comm = comm_of(c);
  c->comm = nullptr; c->comm_rank = 0; c->comm_world = 1;
  NCCLCHK(ncclCommDestroy(comm));
  return NIF_OK;
}

extern "C" int nif_comm_info(nif_ctx* c, int32_t* rank, int32_t* world) {
  if (!c) return fail(NIF_ERR_INVALID, "null");
  if (rank) *rank = c->comm_rank;
  if (world) *world = c->comm_world;
  return NIF_OK;
}

static int dtype_of(int32_t dt, ncclDataType_t* out) {
  switch (dt) {
    case NIF_DT_F32: *out = ncclFloat32; return NIF_OK;
    case NIF_DT_F64: *out = ncclFloat64; return NIF_OK;
    case NIF_DT_I64: *out = ncclInt64; return NIF_OK;
    default: return fail(NIF_ERR_INVALID, "unknown nif_dtype");
  }
}
static int op_of(int32_t op, ncclRedOp_t* out) {
  switch (op) {
    case NIF_OP_SUM: *out = ncclSum; return NIF_OK;
    case NIF_OP_MAX: *out = ncclMax; return NIF_OK;
    case NIF_OP_MIN: *out = ncclMin; return NIF_OK;
    default: return fail(NIF_ERR_INVALID, "unknown nif_redop");
  }
}

// The one collective of the training step.  world == 1 without a communicator: nothing to do.
extern "C" int nif_allreduce_grad(nif_ctx* c) {
  if (!c) return fail(NIF_ERR_INVALID, "null");
  if (!c->comm) return c->comm_world == 1 ? NIF_OK : fail(NIF_ERR_STATE, "no communicator");
  HIPCHK(hipSetDevice(c->dev));
  { const int rct = nif_tail_flush(c); if (rct) return rct; }
  NCCLCHK(ncclAllReduce(c->grad, c->grad, (size_t)(c->P + 1), ncclFloat32, ncclSum, comm_of(c), c->st));
  return NIF_OK;
}

// Self-check of the training collective, for the first run on a new node (bench.py calls it before the warm-up): every rank fills
// THE all-reduce buffer [grad | loss] with rank + 1, nif_allreduce_grad() -- the step's own call, same buffer, count, stream --
// sums it, and first / middle / last element must read world (world + 1) / 2 on every rank.  ranks_seen = the n that solves
// n (n + 1) / 2 = the sum found.  The buffer is zeroed afterwards (the next loss / gradient overwrites it anyway).
__global__ void k_fill_f32(float* __restrict__ p, long n, float v) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] = v;
}
extern "C" int nif_comm_selftest(nif_ctx* c, int32_t* ranks_seen) {
  if (!c || !ranks_seen) return fail(NIF_ERR_INVALID, "null");
  if (!c->grad) return fail(NIF_ERR_STATE, "no gradient buffer");
  HIPCHK(hipSetDevice(c->dev));
  { const int rct = nif_tail_flush(c); if (rct) return rct; }
  const long n = c->P + 1;
  hipLaunchKernelGGL(k_fill_f32, dim3(64), dim3(256), 0, c->st, c->grad, n, (float)(c->comm_rank + 1));
  int rc = nif_allreduce_grad(c); if (rc) return rc;
  float got[3] = {0.f, 0.f, 0.f};
  const long at[3] = {0, n / 2, n - 1};
  for (int i = 0; i < 3; ++i) HIPCHK(hipMemcpyAsync(&got[i], c->grad + at[i], sizeof(float), hipMemcpyDeviceToHost, c->st));
  HIPCHK(hipMemsetAsync(c->grad, 0, sizeof(float) * (size_t)n, c->st));
  HIPCHK(hipStreamSynchronize(c->st));
  const int w = c->comm ? c->comm_world : 1;
  const float want = 0.5f * (float)w * (float)(w + 1);
  *ranks_seen = (int32_t)lroundf(0.5f * (sqrtf(8.f * got[0] + 1.f) - 1.f));
  for (int i = 0; i < 3; ++i)
    if (got[i] != want)
      return fail(NIF_ERR_COMM, "all-reduce self-check: element " + std::to_string(at[i]) + " of [grad | loss] sums to " + std::to_string(got[i]) +
                                    ", expected " + std::to_string(want) + " for " + std::to_string(w) + " ranks");
  return NIF_OK;
}

// same, for the n contexts one process drives: one group call, so RCCL launches all ranks' kernels together
extern "C" int nif_allreduce_grad_multi(nif_ctx** ctxs, int32_t n) {
  if (!ctxs || n < 1) return fail(NIF_ERR_INVALID, "bad argument");
  if (n == 1 && !ctxs[0]->comm) return NIF_OK;
  for (int i = 0; i < n; ++i)
    if (!ctxs[i] || !ctxs[i]->comm || ctxs[i]->comm_world != n) return fail(NIF_ERR_STATE, "contexts are not one nif_comm_init_all group");
  for (int i = 0; i < n; ++i) { const int rct = nif_tail_flush(ctxs[i]); if (rct) return rct; }
  NCCLCHK(ncclGroupStart());
  for (int i = 0; i < n; ++i) {
    nif_ctx* c = ctxs[i];
    ncclResult_t r = ncclAllReduce(c->grad, c->grad, (size_t)(c->P + 1), ncclFloat32, ncclSum, comm_of(c), c->st);
    if (r != ncclSuccess) { (void)ncclGroupEnd(); return fail(NIF_ERR_COMM, std::string("ncclAllReduce: ") + ncclGetErrorString(r)); }
  }
  NCCLCHK(ncclGroupEnd());
  return NIF_OK;
}

// plumbing collective on a caller-owned device buffer (agreed batch sizes of fit, max-over-ranks timing of bench.py)
extern "C" int nif_comm_allreduce(nif_ctx* c, void* dev_buf, int64_t count, int32_t dtype, int32_t op) {
  if (!c || !dev_buf || count < 0) return fail(NIF_ERR_INVALID, "bad argument");
  if (!c->comm) return c->comm_world == 1 ? NIF_OK : fail(NIF_ERR_STATE, "no communicator");
  ncclDataType_t dt; ncclRedOp_t ro;
  int rc = dtype_of(dtype, &dt); if (rc) return rc;
  rc = op_of(op, &ro); if (rc) return rc;
  HIPCHK(hipSetDevice(c->dev));
  NCCLCHK(ncclAllReduce(dev_buf, dev_buf, (size_t)count, dt, ro, comm_of(c), c->st));
  return NIF_OK;
}

// all ranks reach this point, and this context's stream has drained: an all-reduce of one word + a stream sync
extern "C" int nif_comm_barrier(nif_ctx* c) {
  if (!c) return fail(NIF_ERR_INVALID, "null");
  HIPCHK(hipSetDevice(c->dev));
  if (c->comm) NCCLCHK(ncclAllReduce(c->comm_scratch, c->comm_scratch, 1, ncclFloat32, ncclSum, comm_of(c), c->st));
  HIPCHK(hipStreamSynchronize(c->st));
  return NIF_OK;
}

// ---- one process, n GPUs: the sharding train step of SURVEY 8b ---------------------------------------------------
// Rows are split contiguously and evenly (the first B % n contexts take one more); every context computes its shard
// pre-scaled by 1/B, one grouped all-reduce, the identical Adam update on every device.  Host pointers; synchronises.
extern "C" int nif_train_step_multi(nif_ctx** ctxs, int32_t n, const float* xin, const float* y, const float* sw, int64_t B,
                                    const nif_adam* opt, float* loss_out) {
  if (!ctxs || n < 1 || !xin || !y || !opt || B < n) return fail(NIF_ERR_INVALID, "bad argument");
  std::vector<float*> dx(n), dy(n), dsw(n);
  std::vector<int64_t> lo(n + 1);
  const int64_t base = B / n, rem = B % n;
  lo[0] = 0;
  for (int i = 0; i < n; ++i) lo[i + 1] = lo[i] + base + (i < rem ? 1 : 0);
  // One host thread per device stages its shard (the host arrays are pageable: hipMemcpyAsync from them blocks its caller) and
  // enqueues the shard's step behind the copy on the device's own stream -- n copies and n steps in flight at once, instead of
  // device i + 1's copy waiting for the host to be done with device i (r2: a serial loop).  The error text of a failing
  // shard is carried over from its thread (nif_last_error is thread-local).
  int rc = NIF_OK;
  std::vector<int> rcs(n, NIF_OK);
  std::vector<std::string> msgs(n);
  auto shard = [&](int i) {
    nif_ctx* c = ctxs[i];
    const int ncol = c->pi + c->si;
    const int64_t b = lo[i + 1] - lo[i];
    int r = nif_stage_batch(c, xin + lo[i] * ncol, y + lo[i] * c->so, sw ? sw + lo[i] : nullptr, b, &dx[i], &dy[i], &dsw[i]);
    if (r == NIF_OK) r = nif_loss_grad_dev(c, dx[i], dy[i], sw ? dsw[i] : nullptr, b, B);
    rcs[i] = r;
    if (r != NIF_OK) msgs[i] = nif_last_error();
  };
  if (n == 1) shard(0);
  else {
    std::vector<std::thread> th;
    for (int i = 0; i < n; ++i) th.emplace_back(shard, i);
    for (auto& t : th) t.join();
  }
  for (int i = 0; i < n && rc == NIF_OK; ++i)
    if (rcs[i] != NIF_OK) rc = nif_fail(rcs[i], msgs[i]);
  if (rc == NIF_OK && n > 1) rc = nif_allreduce_grad_multi(ctxs, n);
  for (int i = 0; i < n && rc == NIF_OK; ++i) rc = nif_adam_step_dev(ctxs[i], opt);
  if (rc == NIF_OK && loss_out) rc = nif_last_loss(ctxs[0], loss_out);
  for (int i = 0; i < n; ++i) {   // every device idle before the host buffers are handed back
    if (hipSetDevice(ctxs[i]->dev) == hipSuccess) (void)hipStreamSynchronize(ctxs[i]->st);
  }
  return rc;
}
