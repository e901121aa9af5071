// In-library communicator: RCCL over xGMI, one process per GPU, collectives enqueued on the context's own HIP stream
// (SURVEY.md §8b `dre_comm_init`, §8e).  librccl is loaded lazily (dlopen) by the first dre_comm_* call, so a single-GPU user never
// loads it and libdre_hip.so keeps linking against libamdhip64 only.
#pragma once
#include <cstddef>
#include <memory>
#include <string>

#include "common.hpp"

namespace dre {

struct Comm {
    int nranks = 1, rank = 0;
    void* nccl = nullptr;            // ncclComm_t (null for a single rank without RCCL: collectives are local copies)
    int emulate = 0;                 // > 1: ONE process plays this many ranks one after the other (tests of the blocking logic on one GPU)
    size_t bytes_gathered = 0, bytes_reduced = 0, ncalls = 0;
    // HOST transport (dre_comm_init_host): the collectives are staged through pinned host memory and handed to the caller's callbacks — an MPI /
    // gloo / anything channel behind the same comm_allgather entry the engine uses.  For hosts without RCCL and for the tests that run the
    // library's sharded solve on TWO ranks with one GPU (RCCL refuses two ranks on one device).
    int (*host_allgather)(void* user, const void* send, void* recv, size_t bytes_per_rank) = nullptr;   // recv: nranks blocks; send may point into recv
    int (*host_allreduce)(void* user, void* buf, size_t count) = nullptr;                               // sum of doubles, in place
    void* host_user = nullptr;
    volatile int async_failed = 0;   // set by a failed callback of the asynchronous host transport, raised by the next collective call
    void* stage = nullptr;           // pinned
    size_t stage_bytes = 0;
    ~Comm();
};

// 128 bytes (NCCL_UNIQUE_ID_BYTES); rank 0 creates it, the host program hands it to the other ranks (torch.distributed, MPI, a file ...)
void comm_unique_id(void* out128);
std::shared_ptr<Comm> comm_init(Ctx* ctx, int nranks, int rank, const void* id128);
std::shared_ptr<Comm> comm_init_host(Ctx* ctx, int nranks, int rank, int (*ag)(void*, const void*, void*, size_t), int (*ar)(void*, void*, size_t), void* user);
// in-place all-gather: every rank has written its block `rank` of `buf` (nranks blocks of `count` doubles each)
void comm_allgather_inplace(Ctx* ctx, Comm& c, double* buf, size_t count);
void comm_allgather(Ctx* ctx, Comm& c, const double* send, double* recv, size_t count);
void comm_allreduce_sum(Ctx* ctx, Comm& c, double* buf, size_t count);

// Column blocks of an n x k panel over P ranks in whole 16-column tiles (the multifrontal sweeps work on 16-column tiles: a narrower
// block would cost a full tile anyway).  Rank g owns tiles [g*tpr, min((g+1)*tpr, tiles)); the gathered buffer has P*tpr*16 columns.
struct ColBlocks {
    int k = 0, P = 1, tpr = 0;
    ColBlocks(int k_, int P_) : k(k_), P(P_ < 1 ? 1 : P_) { const int tiles = (k + 15) / 16; tpr = (tiles + P - 1) / P; }
    int width() const { return tpr * 16; }                       // columns per block of the gathered buffer
    int c0(int g) const { const int v = g * width(); return v < k ? v : k; }
    int c1(int g) const { const int v = (g + 1) * width(); return v < k ? v : k; }
    int padded() const { return P * width(); }
};

}  // namespace dre
