// api.hip — error plumbing, geometry helper and the op-list runner of libesrgan_hip.so.
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <mutex>

#include "common.h"

namespace {
thread_local char g_err[512] = "";
}

void esr_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int esr_check_launch(const char* what) {
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    esr_set_error("%s: %s", what, hipGetErrorString(e));
    return ESR_ERR_LAUNCH;
  }
  return ESR_OK;
}

// HIP runtime calls on the control path: report instead of ignoring a failure
#define ESR_HIP(call)                                                                   \
  do {                                                                                  \
    const hipError_t e_ = (call);                                                       \
    if (e_ != hipSuccess) {                                                             \
      esr_set_error("%s: %s", #call, hipGetErrorString(e_));                            \
      return ESR_ERR_LAUNCH;                                                            \
    }                                                                                   \
  } while (0)

extern "C" const char* esr_last_error(void) { return g_err; }
extern "C" int esr_abi_version(void) { return 6; }   // 6: round 6 (ESR_OPF_FOLLOW, esr_debug_rdb_wgrad_follow); 5: round 5 (esr_rdb_wgrad.max_workgroups, per-device bookkeeping diagnostics); 4: round 4 (esr_conv.ksplit ..., BN_FIN_APPLY / BN_RESTAT); 3: round 3 (esr_ragan_loss modes, esr_rdb_backward, esr_rdb_wgrad)
extern "C" size_t esr_sizeof_op(void) { return sizeof(esr_op); }

extern "C" void esr_g32_dims(int32_t H, int32_t W, int32_t* Hp, int32_t* Wp) {
  // 1-pixel zero ring + room for the largest tile overhang (rows) and wrap-around reads (cols).
  if (Hp) *Hp = ((H + 31) / 32) * 32 + 6;
  if (Wp) *Wp = ((W + 31) / 32) * 32 + 2;
}

// Side streams for ESR_OPF_SIDE runs: one non-blocking stream + two fork/join event pairs (+ NFREE streams for the
// free runs) per DEVICE, created on first use — the only device resources the library ever owns.  Callers on several
// streams / threads of one device share them: a fork is {record the event on the caller's stream, make the side stream
// wait for it}, done under the state's mutex so that no other caller re-records the event in between (a join event
// re-recorded by another caller only makes this caller wait for more of the in-order side stream: safe).
// ESR_SIDE_PER_STREAM=1 gives every caller stream its own set instead: measured SLOWER on the train step (7.75 vs
// 7.24 ms: more streams in flight, see train.py on stream counts), kept as a knob.
int esr_bookkeeping_device();                    // rdb_fused.hip
namespace {
constexpr int NFREE = 3;   // streams for ESR_OPF_SIDE_FREE runs (independent weight gradients, several at once)
struct SideState { std::mutex mu; hipStream_t owner; hipStream_t stream; hipEvent_t fork[2], join[2]; hipStream_t xs[NFREE]; hipEvent_t xfork, xjoin[NFREE]; SideState* next; };
SideState* side_state(hipStream_t owner) {
  static std::mutex mu;
  static SideState* per_dev[64] = {};
  const int dev = esr_bookkeeping_device();
  static const bool per_stream = [] { const char* e = getenv("ESR_SIDE_PER_STREAM"); return e && atoi(e) != 0; }();
  if (!per_stream) owner = nullptr;              // default: one side state per device
  std::lock_guard<std::mutex> lk(mu);
  for (SideState* s = per_dev[dev]; s; s = s->next)
    if (s->owner == owner) return s;
  SideState* s = new SideState();
  s->owner = owner;
  bool ok = hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking) == hipSuccess;
  for (int k = 0; k < 2 && ok; ++k)
    ok = hipEventCreateWithFlags(&s->fork[k], hipEventDisableTiming) == hipSuccess &&
         hipEventCreateWithFlags(&s->join[k], hipEventDisableTiming) == hipSuccess;
  ok = ok && hipEventCreateWithFlags(&s->xfork, hipEventDisableTiming) == hipSuccess;
  for (int k = 0; k < NFREE && ok; ++k)
    ok = hipStreamCreateWithFlags(&s->xs[k], hipStreamNonBlocking) == hipSuccess &&
         hipEventCreateWithFlags(&s->xjoin[k], hipEventDisableTiming) == hipSuccess;
  if (!ok) { esr_set_error("side stream: creation failed"); delete s; return nullptr; }
  s->next = per_dev[dev];
  per_dev[dev] = s;
  return s;
}
}  // namespace

int esr_rdb_backward_with_follower(const esr_rdb_chain* ch, const esr_rdb_wgrad* wg, hipStream_t stream, hipStream_t side, hipEvent_t fork);   // rdb_fused.hip

// consecutive weight-gradient ops handed to one esr_conv_wgrad_multi call (an RRDB: 3 x 6)
constexpr int ESR_WGRAD_RUN_MAX = 24;

extern "C" int esr_run_ops(const esr_op* ops, int32_t n, esr_stream_t stream) {
  if (!ops || n < 0) { esr_set_error("esr_run_ops: invalid arguments"); return ESR_ERR_INVALID; }
  if (esr_rdb_check_abort()) {
    esr_set_error("esr_run_ops: an earlier fused-chain launch aborted (a tile waited > 1 s for its neighbours: CUs held by other work?) — its results are invalid");
    return ESR_ERR_LAUNCH;
  }
  int nside = 0;        // side runs launched by this call
  bool joined = true;   // main stream already waits for the last side run
  int nfree = 0;        // ESR_OPF_SIDE_FREE runs launched by this call (joined at the unpermute / the end)
  auto join_free = [&]() -> int {
    if (nfree > 0) {
      SideState* ss = side_state((hipStream_t)stream);
      for (int q = 0; q < NFREE && q < nfree; ++q) ESR_HIP(hipStreamWaitEvent((hipStream_t)stream, ss->xjoin[q], 0));
      nfree = 0;
    }
    return ESR_OK;
  };
  for (int i = 0; i < n; ++i) {
    int rc;
    switch (ops[i].kind) {
      case ESR_OP_CONV: rc = esr_conv_forward(&ops[i].u.conv, stream); break;
      case ESR_OP_PACK: rc = esr_pack_conv_weights(&ops[i].u.pack, stream); break;
      case ESR_OP_LAYOUT: rc = esr_convert_layout(&ops[i].u.layout, stream); break;
      case ESR_OP_NOISE_FILL: rc = esr_fill_noise(&ops[i].u.noise_fill, stream); break;
      case ESR_OP_WGRAD: {
        // consecutive weight-gradient ops are independent by construction (disjoint dW blocks, read
        // only g / saved inputs): hand the run to the batched launcher
        const int side = ops[i].flags & ESR_OPF_SIDE;
        int m = 1;
        while (i + m < n && m < ESR_WGRAD_RUN_MAX && ops[i + m].kind == ESR_OP_WGRAD && (ops[i + m].flags & ESR_OPF_SIDE) == side) ++m;
        esr_wgrad run[ESR_WGRAD_RUN_MAX];
        for (int k = 0; k < m; ++k) run[k] = ops[i + k].u.wgrad;
        if (!side) {
          rc = m == 1 ? esr_conv_wgrad(&run[0], stream) : esr_conv_wgrad_multi(run, m, stream);
        } else if (ops[i].flags & ESR_OPF_SIDE_FREE) {
          // what this run reads is never overwritten inside the list and its partial arena is its own: no wait
          // for earlier runs, round-robin over NFREE streams (latency-bound launches overlap each other)
          SideState* ss = side_state((hipStream_t)stream);
          if (!ss) return ESR_ERR_LAUNCH;
          const int q = nfree % NFREE;
          {
            std::lock_guard<std::mutex> lk(ss->mu);
            ESR_HIP(hipEventRecord(ss->xfork, (hipStream_t)stream));
            ESR_HIP(hipStreamWaitEvent(ss->xs[q], ss->xfork, 0));
          }
          rc = esr_conv_wgrad_multi(run, m, (esr_stream_t)ss->xs[q]);
          ESR_HIP(hipEventRecord(ss->xjoin[q], ss->xs[q]));
          ++nfree;
        } else {
          // fork: side stream waits for everything enqueued so far; the previous side run must be done
          // before anything after this point (its inputs may be overwritten from here on)
          SideState* ss = side_state((hipStream_t)stream);
          if (!ss) return ESR_ERR_LAUNCH;
          hipStream_t main_st = (hipStream_t)stream;
          if (nside > 0) ESR_HIP(hipStreamWaitEvent(main_st, ss->join[(nside - 1) & 1], 0));
          {
            std::lock_guard<std::mutex> lk(ss->mu);
            ESR_HIP(hipEventRecord(ss->fork[nside & 1], main_st));
            ESR_HIP(hipStreamWaitEvent(ss->stream, ss->fork[nside & 1], 0));
          }
          rc = esr_conv_wgrad_multi(run, m, (esr_stream_t)ss->stream);
          ESR_HIP(hipEventRecord(ss->join[nside & 1], ss->stream));
          ++nside;
          joined = false;
        }
        if (rc == ESR_OK) i += m - 1;
        break;
      }
      case ESR_OP_BN: rc = esr_batchnorm(&ops[i].u.bn, stream); break;
      case ESR_OP_POOL: rc = esr_maxpool2(&ops[i].u.pool, stream); break;
      case ESR_OP_LINEAR: rc = esr_linear_op(&ops[i].u.linear, stream); break;
      case ESR_OP_UNPERMUTE:
        if ((rc = join_free()) != ESR_OK) break;
        if (nside > 0 && !joined) { ESR_HIP(hipStreamWaitEvent((hipStream_t)stream, side_state((hipStream_t)stream)->join[(nside - 1) & 1], 0)); joined = true; }
        rc = esr_grad_unpermute(&ops[i].u.unpermute, stream);
        break;
      case ESR_OP_PACK_BATCH: rc = esr_pack_conv_weights_batch(&ops[i].u.pack_batch, stream); break;
      case ESR_OP_RDB_CHAIN: rc = esr_rdb_forward(&ops[i].u.rdb_chain, stream); break;
      case ESR_OP_RDB_CHAIN_BWD: {
        // a follower pass (ESR_OPF_FOLLOW on the ESR_OP_RDB_WGRAD op right behind): chain and weight gradients launched
        // together — the pass on the side stream, behind the clearing of the chain's flags, one block behind the chain
        if (i + 1 < n && ops[i + 1].kind == ESR_OP_RDB_WGRAD && (ops[i + 1].flags & ESR_OPF_FOLLOW)) {
          SideState* ss = side_state((hipStream_t)stream);
          if (!ss) return ESR_ERR_LAUNCH;
          hipStream_t main_st = (hipStream_t)stream;
          if (nside > 0) ESR_HIP(hipStreamWaitEvent(main_st, ss->join[(nside - 1) & 1], 0));
          {
            std::lock_guard<std::mutex> lk(ss->mu);
            rc = esr_rdb_backward_with_follower(&ops[i].u.rdb_chain, &ops[i + 1].u.rdb_wgrad, main_st, ss->stream, ss->fork[nside & 1]);
          }
          if (rc == ESR_OK) {
            ESR_HIP(hipEventRecord(ss->join[nside & 1], ss->stream));
            ++nside;
            joined = false;
            ++i;
          }
          break;
        }
        rc = esr_rdb_backward(&ops[i].u.rdb_chain, stream);
        break;
      }
      case ESR_OP_FRAG_GATHER: rc = esr_gather_fragments(&ops[i].u.frag_gather, stream); break;
      case ESR_OP_RDB_WGRAD: {
        // dense-block weight gradients: like a run of wgrad ops — on the side stream when flagged ESR_OPF_SIDE
        if (!(ops[i].flags & ESR_OPF_SIDE)) { rc = esr_rdb_wgrad_run(&ops[i].u.rdb_wgrad, stream); break; }
        SideState* ss = side_state((hipStream_t)stream);
        if (!ss) return ESR_ERR_LAUNCH;
        hipStream_t main_st = (hipStream_t)stream;
        if (nside > 0) ESR_HIP(hipStreamWaitEvent(main_st, ss->join[(nside - 1) & 1], 0));
        {
          std::lock_guard<std::mutex> lk(ss->mu);
          ESR_HIP(hipEventRecord(ss->fork[nside & 1], main_st));
          ESR_HIP(hipStreamWaitEvent(ss->stream, ss->fork[nside & 1], 0));
        }
        rc = esr_rdb_wgrad_run(&ops[i].u.rdb_wgrad, (esr_stream_t)ss->stream);
        ESR_HIP(hipEventRecord(ss->join[nside & 1], ss->stream));
        ++nside;
        joined = false;
        break;
      }
      default: esr_set_error("esr_run_ops: op %d has unknown kind %d", i, ops[i].kind); return ESR_ERR_INVALID;
    }
    if (rc != ESR_OK) {
      char tmp[400];
      snprintf(tmp, sizeof(tmp), "%s", esr_last_error());
      esr_set_error("op %d (kind %d): %s", i, ops[i].kind, tmp);
      if (nside > 0 && !joined) (void)hipStreamWaitEvent((hipStream_t)stream, side_state((hipStream_t)stream)->join[(nside - 1) & 1], 0);
      (void)join_free();
      return rc;
    }
  }
  if (nside > 0 && !joined) ESR_HIP(hipStreamWaitEvent((hipStream_t)stream, side_state((hipStream_t)stream)->join[(nside - 1) & 1], 0));
  return join_free();
}

extern "C" int64_t esr_wgrad_run_partial_elems(const esr_wgrad* items, int32_t n);
extern "C" int64_t esr_wgrad_workspace_elems(const esr_op* ops, int32_t n) {
  // the same runs esr_run_ops forms: consecutive wgrad ops with one side flag, at most 16
  int64_t need = 0;
  for (int i = 0; ops && i < n; ++i) {
    if (ops[i].kind != ESR_OP_WGRAD) continue;
    const int side = ops[i].flags & ESR_OPF_SIDE;
    int m = 1;
    while (i + m < n && m < ESR_WGRAD_RUN_MAX && ops[i + m].kind == ESR_OP_WGRAD && (ops[i + m].flags & ESR_OPF_SIDE) == side) ++m;
    esr_wgrad run[ESR_WGRAD_RUN_MAX];
    for (int k = 0; k < m; ++k) run[k] = ops[i + k].u.wgrad;
    const int64_t e = esr_wgrad_run_partial_elems(run, m);
    if (e > need) need = e;
    i += m - 1;
  }
  return need;     // (ESR_OPF_SIDE_FREE runs: the caller gives each its own region of this size or of its own need)
}

// ------------------------------------------------------------------------------------------------
// hipGraph replay
// ------------------------------------------------------------------------------------------------
struct esr_graph_s { hipGraph_t graph; hipGraphExec_t exec; bool has_chain; };
int esr_chain_graph_before(hipStream_t st);             // rdb_fused.hip
void esr_chain_graph_after(hipStream_t st, int cus);

namespace {
hipStream_t capture_stream() {
  static std::mutex mu;
  static hipStream_t per_dev[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  std::lock_guard<std::mutex> lk(mu);
  if (!per_dev[dev] && hipStreamCreateWithFlags(&per_dev[dev], hipStreamNonBlocking) != hipSuccess) return nullptr;
  return per_dev[dev];
}
}  // namespace

extern "C" int esr_graph_create(const esr_op* ops, int32_t n, esr_graph_t* out) {
  if (!ops || n <= 0 || !out) { esr_set_error("esr_graph_create: invalid arguments"); return ESR_ERR_INVALID; }
  hipStream_t cs = capture_stream();
  if (!cs) { esr_set_error("esr_graph_create: no capture stream"); return ESR_ERR_LAUNCH; }
  static std::mutex cap_mu;                    // one capture at a time on the shared capture stream
  std::lock_guard<std::mutex> lk(cap_mu);
  hipError_t e = hipStreamBeginCapture(cs, hipStreamCaptureModeRelaxed);
  if (e != hipSuccess) { esr_set_error("esr_graph_create: begin capture: %s", hipGetErrorString(e)); return ESR_ERR_LAUNCH; }
  const int rc = esr_run_ops(ops, n, (esr_stream_t)cs);
  hipGraph_t g = nullptr;
  e = hipStreamEndCapture(cs, &g);
  if (rc != ESR_OK) { if (g) (void)hipGraphDestroy(g); return rc; }
  if (e != hipSuccess || !g) { esr_set_error("esr_graph_create: end capture: %s", hipGetErrorString(e)); return ESR_ERR_LAUNCH; }
  hipGraphExec_t x = nullptr;
  e = hipGraphInstantiate(&x, g, nullptr, nullptr, 0);
  if (e != hipSuccess) {
    (void)hipGraphDestroy(g);
    esr_set_error("esr_graph_create: instantiate: %s", hipGetErrorString(e));
    return ESR_ERR_LAUNCH;
  }
  esr_graph_s* h = new esr_graph_s();
  h->graph = g;
  h->exec = x;
  h->has_chain = false;
  for (int i = 0; i < n; ++i)
    if (ops[i].kind == ESR_OP_RDB_CHAIN || ops[i].kind == ESR_OP_RDB_CHAIN_BWD) h->has_chain = true;
  *out = h;
  return ESR_OK;
}

extern "C" int esr_graph_launch(esr_graph_t g, esr_stream_t stream) {
  if (!g || !g->exec) { esr_set_error("esr_graph_launch: invalid graph"); return ESR_ERR_INVALID; }
  int cus = 0;
  if (g->has_chain) {
    // chain nodes replayed from a graph bypass chain_launch's bookkeeping: report an earlier abort here, and order the
    // replay against chains in flight on other streams as one whole-GPU chain launch
    if (esr_rdb_check_abort()) {
      esr_set_error("esr_graph_launch: an earlier fused-chain launch aborted (a tile waited > 1 s for its neighbours: CUs held by other work?) — its results are invalid");
      return ESR_ERR_LAUNCH;
    }
    cus = esr_chain_graph_before((hipStream_t)stream);
  }
  const hipError_t e = hipGraphLaunch(g->exec, (hipStream_t)stream);
  if (e != hipSuccess) { esr_set_error("esr_graph_launch: %s", hipGetErrorString(e)); return ESR_ERR_LAUNCH; }
  if (g->has_chain) esr_chain_graph_after((hipStream_t)stream, cus);
  return ESR_OK;
}

extern "C" int esr_graph_destroy(esr_graph_t g) {
  if (!g) return ESR_OK;
  if (g->exec) (void)hipGraphExecDestroy(g->exec);
  if (g->graph) (void)hipGraphDestroy(g->graph);
  delete g;
  return ESR_OK;
}

extern "C" int esr_run_ops_timed(const esr_op* ops, int32_t n, esr_stream_t stream, float* ms_out) {
  if (!ops || n <= 0 || !ms_out) { esr_set_error("esr_run_ops_timed: invalid arguments"); return ESR_ERR_INVALID; }
  hipStream_t st = (hipStream_t)stream;
  hipEvent_t* ev = new hipEvent_t[n + 1];
  for (int i = 0; i <= n; ++i) (void)hipEventCreate(&ev[i]);
  int rc = ESR_OK;
  (void)hipEventRecord(ev[0], st);
  for (int i = 0; i < n && rc == ESR_OK; ++i) {
    rc = esr_run_ops(&ops[i], 1, stream);
    (void)hipEventRecord(ev[i + 1], st);
  }
  if (hipStreamSynchronize(st) != hipSuccess && rc == ESR_OK) { esr_set_error("esr_run_ops_timed: stream sync failed"); rc = ESR_ERR_LAUNCH; }
  if (rc == ESR_OK)
    for (int i = 0; i < n; ++i) (void)hipEventElapsedTime(&ms_out[i], ev[i], ev[i + 1]);
  for (int i = 0; i <= n; ++i) (void)hipEventDestroy(ev[i]);
  delete[] ev;
  return rc;
}
