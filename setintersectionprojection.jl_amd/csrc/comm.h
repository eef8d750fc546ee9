// Collectives of the sharded solve.  The reference's parallel mode moves data between Julia workers through
// DistributedArrays / @spawnat (src/PARSDMM.jl:114-131, src/rhs_compose.jl:17-20, src/update_y_l_parallel.jl:6-90); here one
// process drives one GPU and the engine itself enqueues the collectives on its HIP streams:
//   * RcclComm     -- RCCL (librccl.so.1, looked up with dlopen so that a single-GPU user never loads it) over xGMI: the
//                     production path, communicator built from an ncclUniqueId the host side distributes;
//   * CallbackComm -- the same three operations supplied by the caller as C function pointers (sipx_comm in sipx.h): what
//                     the test harness uses to run several ranks on one GPU over gloo, and what a host that already owns a
//                     communicator (torch.distributed, MPI) would plug in.
// All operations are IN PLACE on a buffer of world * chunk elements and are enqueued on the given stream.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

#include "../../include/sipx.h"

namespace sipx {

struct Comm {
  int world = 1, rank = 0;
  virtual ~Comm() {}
  virtual const char* kind() const = 0;
  // what the communicator itself reports: ranks it spans, this rank's number in it, its library version ("" if none)
  virtual void info(int* nranks, int* rank_out, char* version, size_t version_len) const = 0;
  // buf[0 .. count) <- sum over ranks, identical bits on every rank
  virtual void allreduce_sum(void* buf, size_t count, int dtype, hipStream_t s) = 0;
  // buf[rank*chunk .. (rank+1)*chunk) <- sum over ranks of that range; the rest of buf is scratch afterwards
  virtual void reduce_scatter_sum(void* buf, size_t chunk, int dtype, hipStream_t s) = 0;
  // buf[r*chunk .. (r+1)*chunk) <- rank r's range, for every r
  virtual void allgather(void* buf, size_t chunk, int dtype, hipStream_t s) = 0;
  // neighbour exchange of the slab CG: send `count` elements to / receive from rank `prev` and rank `next` (-1: no such
  // neighbour); the four buffers are distinct device ranges
  virtual void halo_exchange(const void* send_prev, void* recv_prev, int prev, const void* send_next, void* recv_next, int next,
                             size_t count, int dtype, hipStream_t s) = 0;
  // the two at once, for steps whose all-reduce and neighbour exchange do not depend on each other (CG: ||r||^2 partials and
  // the boundary planes of r): ONE call where the communicator can group them (RCCL: one ncclGroup, one launch), else in turn
  virtual void allreduce_with_halo(void* buf, size_t count, int red_dtype, const void* send_prev, void* recv_prev, int prev,
                                   const void* send_next, void* recv_next, int next, size_t hcount, int hdtype, hipStream_t s) {
    allreduce_sum(buf, count, red_dtype, s);
    halo_exchange(send_prev, recv_prev, prev, send_next, recv_next, next, hcount, hdtype, s);
  }
  // in place on world * chunk elements: rank r receives buf[r*chunk .. (r+1)*chunk) of rank `root` (scatter), or rank `root`
  // receives that range from every rank r (gather); the root's own range stays where it is
  virtual void scatter(void* buf, size_t chunk, int dtype, int root, hipStream_t s) = 0;
  virtual void gather(void* buf, size_t chunk, int dtype, int root, hipStream_t s) = 0;
  // recv[r*chunk .. (r+1)*chunk) <- send[rank*chunk .. (rank+1)*chunk) of rank r, for every r (the transposition of the slab-decomposed
  // DFT, round 5).  send, recv and tmp are distinct buffers of world * chunk elements; tmp is scratch for communicators without
  // an all-to-all of their own: the default strings it together from one scatter per root (callbacks: sipx_comm has none).
  virtual void alltoall(const void* send, void* recv, void* tmp, size_t chunk, int dtype, hipStream_t s) {
    const size_t bytes = chunk * (dtype == SIPX_F64 ? 8 : 4);
    for (int root = 0; root < world; ++root) {
      if (rank == root) (void)hipMemcpyAsync(tmp, send, bytes * (size_t)world, hipMemcpyDeviceToDevice, s);
      scatter(tmp, chunk, dtype, root, s);
      (void)hipMemcpyAsync(static_cast<char*>(recv) + (size_t)root * bytes, static_cast<char*>(tmp) + (size_t)rank * bytes, bytes,
                           hipMemcpyDeviceToDevice, s);
    }
  }
};

// Forwards every operation to the communicator it wraps and counts the calls by kind (the engine reports them in its statistics:
// how many collectives an iteration of a decomposition issues is the first thing its scaling depends on).
struct CountingComm : Comm {
  Comm* inner;
  long long n_allreduce = 0, n_reduce_scatter = 0, n_allgather = 0, n_halo = 0, n_grouped = 0, n_fan = 0;
  explicit CountingComm(Comm* c) : inner(c) { world = c->world; rank = c->rank; }
  ~CountingComm() override { delete inner; }
  const char* kind() const override { return inner->kind(); }
  void info(int* nranks, int* rank_out, char* version, size_t version_len) const override { inner->info(nranks, rank_out, version, version_len); }
  void allreduce_sum(void* buf, size_t count, int dtype, hipStream_t s) override { ++n_allreduce; inner->allreduce_sum(buf, count, dtype, s); }
  void reduce_scatter_sum(void* buf, size_t chunk, int dtype, hipStream_t s) override { ++n_reduce_scatter; inner->reduce_scatter_sum(buf, chunk, dtype, s); }
  void allgather(void* buf, size_t chunk, int dtype, hipStream_t s) override { ++n_allgather; inner->allgather(buf, chunk, dtype, s); }
  void halo_exchange(const void* send_prev, void* recv_prev, int prev, const void* send_next, void* recv_next, int next, size_t count, int dtype,
                     hipStream_t s) override {
    ++n_halo;
    inner->halo_exchange(send_prev, recv_prev, prev, send_next, recv_next, next, count, dtype, s);
  }
  void allreduce_with_halo(void* buf, size_t count, int red_dtype, const void* send_prev, void* recv_prev, int prev, const void* send_next,
                           void* recv_next, int next, size_t hcount, int hdtype, hipStream_t s) override {
    ++n_grouped;
    inner->allreduce_with_halo(buf, count, red_dtype, send_prev, recv_prev, prev, send_next, recv_next, next, hcount, hdtype, s);
  }
  void scatter(void* buf, size_t chunk, int dtype, int root, hipStream_t s) override { ++n_fan; inner->scatter(buf, chunk, dtype, root, s); }
  void gather(void* buf, size_t chunk, int dtype, int root, hipStream_t s) override { ++n_fan; inner->gather(buf, chunk, dtype, root, s); }
  void alltoall(const void* send, void* recv, void* tmp, size_t chunk, int dtype, hipStream_t s) override {
    ++n_alltoall;
    inner->alltoall(send, recv, tmp, chunk, dtype, s);
  }
  long long n_alltoall = 0;
};

Comm* make_rccl_comm(const void* unique_id, int world, int rank);
void rccl_unique_id(void* out128);
Comm* make_callback_comm(const sipx_comm* cb);

}  // namespace sipx
