// Collectives of the sharded solve: RCCL (native, over xGMI) and caller-supplied callbacks.  See comm.h.
#include "comm.h"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <stdexcept>
#include <string>

namespace sipx {

namespace {

// librccl.so.1 is looked up at run time: the dynamic loader hands back the copy that is already mapped when the host
// process has one (PyTorch ships its own under the same soname, next to the HIP runtime it was built against), so the
// engine never mixes two RCCL builds; a process that never shards never loads the library at all.
struct RcclApi {
  void* h = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclReduceScatter) ReduceScatter = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclSend) Send = nullptr;
  decltype(&ncclRecv) Recv = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  decltype(&ncclCommCount) CommCount = nullptr;
  decltype(&ncclCommUserRank) CommUserRank = nullptr;
  decltype(&ncclGetVersion) GetVersion = nullptr;
};

const RcclApi& rccl() {
  static RcclApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      api.h = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (api.h) break;
    }
    if (!api.h) return;
    auto sym = [&](auto& fn, const char* n) { fn = reinterpret_cast<std::decay_t<decltype(fn)>>(dlsym(api.h, n)); };
    sym(api.GetUniqueId, "ncclGetUniqueId");
    sym(api.CommInitRank, "ncclCommInitRank");
    sym(api.CommDestroy, "ncclCommDestroy");
    sym(api.AllReduce, "ncclAllReduce");
    sym(api.ReduceScatter, "ncclReduceScatter");
    sym(api.AllGather, "ncclAllGather");
    sym(api.Send, "ncclSend");
    sym(api.Recv, "ncclRecv");
    sym(api.GroupStart, "ncclGroupStart");
    sym(api.GroupEnd, "ncclGroupEnd");
    sym(api.GetErrorString, "ncclGetErrorString");
    sym(api.CommCount, "ncclCommCount");
    sym(api.CommUserRank, "ncclCommUserRank");
    sym(api.GetVersion, "ncclGetVersion");
  });
  if (!api.h || !api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.AllReduce || !api.ReduceScatter ||
      !api.AllGather || !api.Send || !api.Recv || !api.GroupStart || !api.GroupEnd)
    throw std::runtime_error("libsipx: librccl.so.1 could not be loaded -- the sharded solve needs RCCL");
  return api;
}

void nccl_check(ncclResult_t r, const char* what) {
  if (r == ncclSuccess) return;
  const RcclApi& a = rccl();
  throw std::runtime_error(std::string(what) + ": " + (a.GetErrorString ? a.GetErrorString(r) : "RCCL error"));
}

inline ncclDataType_t nccl_type(int dtype) { return dtype == SIPX_F64 ? ncclDouble : ncclFloat; }
inline size_t type_size(int dtype) { return dtype == SIPX_F64 ? 8 : 4; }

class RcclComm : public Comm {
 public:
  RcclComm(const void* id, int world_, int rank_) {
    world = world_;
    rank = rank_;
    if (world < 1 || rank < 0 || rank >= world) throw std::runtime_error("RCCL communicator: rank / world out of range");
    ncclUniqueId uid;
    std::memcpy(&uid, id, sizeof(uid));
    nccl_check(rccl().CommInitRank(&comm_, world, uid, rank), "ncclCommInitRank");
    // Whether an all-reduce and send/recv pairs travel in ONE ncclGroup is decided HERE, once, from facts every rank shares --
    // the library's version (collectives and point-to-point calls may share a group from NCCL 2.8 on; RCCL follows NCCL's
    // numbering) and SIPX_COMM_GROUP -- never from the outcome of a call inside a solve: a failed group leaves the
    // communicator in an error state, part of it may have been launched, and a rank without neighbours would not even see the
    // failure.  Any RCCL error during a solve is fatal (nccl_check).
    int v = 0;
    if (rccl().GetVersion) (void)rccl().GetVersion(&v);
    const char* e = std::getenv("SIPX_COMM_GROUP");
    grouped_ = v >= 20800 && !(e && e[0] == '0');
  }
  ~RcclComm() override {
    if (comm_) (void)rccl().CommDestroy(comm_);
  }
  const char* kind() const override { return "rccl"; }
  // asked of RCCL itself (ncclCommCount / ncclCommUserRank / ncclGetVersion), not echoed from the constructor's arguments:
  // "did RCCL see N ranks" has to be answerable from a bench line
  void info(int* nranks, int* rank_out, char* version, size_t version_len) const override {
    const RcclApi& a = rccl();
    int n = -1, r = -1, v = 0;
    if (a.CommCount) nccl_check(a.CommCount(comm_, &n), "ncclCommCount");
    if (a.CommUserRank) nccl_check(a.CommUserRank(comm_, &r), "ncclCommUserRank");
    if (a.GetVersion) nccl_check(a.GetVersion(&v), "ncclGetVersion");
    if (nranks) *nranks = n;
    if (rank_out) *rank_out = r;
    if (version && version_len) {
      // NCCL_VERSION_CODE: major * 10000 + minor * 100 + patch from 2.9 on
      std::snprintf(version, version_len, "rccl %d.%d.%d", v / 10000, (v / 100) % 100, v % 100);
    }
  }
  void allreduce_sum(void* buf, size_t count, int dtype, hipStream_t s) override {
    nccl_check(rccl().AllReduce(buf, buf, count, nccl_type(dtype), ncclSum, comm_, s), "ncclAllReduce");
  }
  void reduce_scatter_sum(void* buf, size_t chunk, int dtype, hipStream_t s) override {
    // in place: recvbuff == sendbuff + rank * recvcount
    char* b = static_cast<char*>(buf);
    nccl_check(rccl().ReduceScatter(b, b + (size_t)rank * chunk * type_size(dtype), chunk, nccl_type(dtype), ncclSum, comm_, s),
               "ncclReduceScatter");
  }
  void allgather(void* buf, size_t chunk, int dtype, hipStream_t s) override {
    // in place: sendbuff == recvbuff + rank * sendcount
    char* b = static_cast<char*>(buf);
    nccl_check(rccl().AllGather(b + (size_t)rank * chunk * type_size(dtype), b, chunk, nccl_type(dtype), comm_, s), "ncclAllGather");
  }
  void halo_exchange(const void* send_prev, void* recv_prev, int prev, const void* send_next, void* recv_next, int next,
                     size_t count, int dtype, hipStream_t s) override {
    if (prev < 0 && next < 0) return;
    const RcclApi& a = rccl();
    // one group: the four transfers (two per neighbour) progress together over the direct xGMI links
    nccl_check(a.GroupStart(), "ncclGroupStart");
    if (prev >= 0) {
      nccl_check(a.Send(send_prev, count, nccl_type(dtype), prev, comm_, s), "ncclSend");
      nccl_check(a.Recv(recv_prev, count, nccl_type(dtype), prev, comm_, s), "ncclRecv");
    }
    if (next >= 0) {
      nccl_check(a.Send(send_next, count, nccl_type(dtype), next, comm_, s), "ncclSend");
      nccl_check(a.Recv(recv_next, count, nccl_type(dtype), next, comm_, s), "ncclRecv");
    }
    nccl_check(a.GroupEnd(), "ncclGroupEnd");
  }
  // all-reduce + neighbour exchange in ONE group (RCCL's planner puts collective and point-to-point work of a group into one
  // launch where the channels allow); SIPX_COMM_GROUP=0 issues them one after the other
  void allreduce_with_halo(void* buf, size_t count, int red_dtype, const void* send_prev, void* recv_prev, int prev,
                           const void* send_next, void* recv_next, int next, size_t hcount, int hdtype, hipStream_t s) override {
    if (!grouped_ || world == 1) {
      Comm::allreduce_with_halo(buf, count, red_dtype, send_prev, recv_prev, prev, send_next, recv_next, next, hcount, hdtype, s);
      return;
    }
    const RcclApi& a = rccl();
    nccl_check(a.GroupStart(), "ncclGroupStart");
    nccl_check(a.AllReduce(buf, buf, count, nccl_type(red_dtype), ncclSum, comm_, s), "ncclAllReduce (grouped)");
    if (prev >= 0) {
      nccl_check(a.Send(send_prev, hcount, nccl_type(hdtype), prev, comm_, s), "ncclSend (grouped)");
      nccl_check(a.Recv(recv_prev, hcount, nccl_type(hdtype), prev, comm_, s), "ncclRecv (grouped)");
    }
    if (next >= 0) {
      nccl_check(a.Send(send_next, hcount, nccl_type(hdtype), next, comm_, s), "ncclSend (grouped)");
      nccl_check(a.Recv(recv_next, hcount, nccl_type(hdtype), next, comm_, s), "ncclRecv (grouped)");
    }
    nccl_check(a.GroupEnd(), "ncclGroupEnd (all-reduce + send/recv in one group; SIPX_COMM_GROUP=0 issues them separately)");
  }
  // one group of point-to-point transfers: the root talks to its seven peers over seven links at once
  void scatter(void* buf, size_t chunk, int dtype, int root, hipStream_t s) override { fan(buf, chunk, dtype, root, s, true); }
  void gather(void* buf, size_t chunk, int dtype, int root, hipStream_t s) override { fan(buf, chunk, dtype, root, s, false); }
  // one group of world sends and world receives (the pattern of ncclAllToAll): every pair of ranks over its own xGMI link
  void alltoall(const void* send, void* recv, void* tmp, size_t chunk, int dtype, hipStream_t s) override {
    (void)tmp;
    const RcclApi& a = rccl();
    const size_t bytes = chunk * type_size(dtype);
    if (world == 1) {
      if (hipMemcpyAsync(recv, send, bytes, hipMemcpyDeviceToDevice, s) != hipSuccess) throw std::runtime_error("all-to-all of one rank: copy failed");
      return;
    }
    const char* sb = static_cast<const char*>(send);
    char* rb = static_cast<char*>(recv);
    // (the rank's own range: a copy on the stream, not a send to itself)
    if (hipMemcpyAsync(rb + (size_t)rank * bytes, sb + (size_t)rank * bytes, bytes, hipMemcpyDeviceToDevice, s) != hipSuccess)
      throw std::runtime_error("all-to-all: the copy of the rank's own range failed");
    nccl_check(a.GroupStart(), "ncclGroupStart");
    for (int p = 0; p < world; ++p) {
      if (p == rank) continue;
      nccl_check(a.Send(sb + (size_t)p * bytes, chunk, nccl_type(dtype), p, comm_, s), "ncclSend (all-to-all)");
      nccl_check(a.Recv(rb + (size_t)p * bytes, chunk, nccl_type(dtype), p, comm_, s), "ncclRecv (all-to-all)");
    }
    nccl_check(a.GroupEnd(), "ncclGroupEnd (all-to-all)");
  }

 private:
  void fan(void* buf, size_t chunk, int dtype, int root, hipStream_t s, bool out) {
    if (world == 1) return;
    const RcclApi& a = rccl();
    char* b = static_cast<char*>(buf);
    const size_t bytes = chunk * type_size(dtype);
    nccl_check(a.GroupStart(), "ncclGroupStart");
    if (rank == root) {
      for (int p = 0; p < world; ++p) {
        if (p == root) continue;
        if (out) nccl_check(a.Send(b + (size_t)p * bytes, chunk, nccl_type(dtype), p, comm_, s), "ncclSend");
        else nccl_check(a.Recv(b + (size_t)p * bytes, chunk, nccl_type(dtype), p, comm_, s), "ncclRecv");
      }
    } else {
      if (out) nccl_check(a.Recv(b + (size_t)rank * bytes, chunk, nccl_type(dtype), root, comm_, s), "ncclRecv");
      else nccl_check(a.Send(b + (size_t)rank * bytes, chunk, nccl_type(dtype), root, comm_, s), "ncclSend");
    }
    nccl_check(a.GroupEnd(), "ncclGroupEnd");
  }
  ncclComm_t comm_ = nullptr;
  bool grouped_ = true;       // all-reduce + neighbour exchange in one ncclGroup: fixed at construction, the same on every rank
};

class CallbackComm : public Comm {
 public:
  explicit CallbackComm(const sipx_comm* cb) : cb_(*cb) {
    world = cb->world;
    rank = cb->rank;
    if (world < 1 || rank < 0 || rank >= world) throw std::runtime_error("sipx_comm: rank / world out of range");
    if (!cb->allreduce_sum || !cb->reduce_scatter_sum || !cb->allgather || !cb->halo_exchange || !cb->scatter || !cb->gather)
      throw std::runtime_error("sipx_comm: every operation must be supplied");
  }
  const char* kind() const override { return "callback"; }
  void info(int* nranks, int* rank_out, char* version, size_t version_len) const override {
    if (nranks) *nranks = world;
    if (rank_out) *rank_out = rank;
    if (version && version_len) std::snprintf(version, version_len, "callbacks (sipx_set_comm)");
  }
  void allreduce_sum(void* buf, size_t count, int dtype, hipStream_t s) override {
    chk(cb_.allreduce_sum(cb_.user, buf, (int64_t)count, dtype, (void*)s), "allreduce_sum");
  }
  void reduce_scatter_sum(void* buf, size_t chunk, int dtype, hipStream_t s) override {
    chk(cb_.reduce_scatter_sum(cb_.user, buf, (int64_t)chunk, dtype, (void*)s), "reduce_scatter_sum");
  }
  void allgather(void* buf, size_t chunk, int dtype, hipStream_t s) override {
    chk(cb_.allgather(cb_.user, buf, (int64_t)chunk, dtype, (void*)s), "allgather");
  }
  void halo_exchange(const void* send_prev, void* recv_prev, int prev, const void* send_next, void* recv_next, int next,
                     size_t count, int dtype, hipStream_t s) override {
    if (prev < 0 && next < 0) return;
    chk(cb_.halo_exchange(cb_.user, send_prev, recv_prev, prev, send_next, recv_next, next, (int64_t)count, dtype, (void*)s),
        "halo_exchange");
  }
  void scatter(void* buf, size_t chunk, int dtype, int root, hipStream_t s) override {
    chk(cb_.scatter(cb_.user, buf, (int64_t)chunk, dtype, root, (void*)s), "scatter");
  }
  void gather(void* buf, size_t chunk, int dtype, int root, hipStream_t s) override {
    chk(cb_.gather(cb_.user, buf, (int64_t)chunk, dtype, root, (void*)s), "gather");
  }

 private:
  static void chk(int rc, const char* what) {
    if (rc != 0) throw std::runtime_error(std::string("sipx_comm callback failed: ") + what);
  }
  sipx_comm cb_;
};

}  // namespace

Comm* make_rccl_comm(const void* unique_id, int world, int rank) { return new RcclComm(unique_id, world, rank); }
void rccl_unique_id(void* out128) {
  ncclUniqueId id;
  nccl_check(rccl().GetUniqueId(&id), "ncclGetUniqueId");
  std::memcpy(out128, &id, sizeof(id));
}
Comm* make_callback_comm(const sipx_comm* cb) { return new CallbackComm(cb); }

}  // namespace sipx
