// superviseddescent/hip_backend.hpp -- RAII handle over the C-ABI (include/sdm.h) for the header layer.
//
// Status codes of the C-ABI are turned back into the exceptions the reference's headers throw
// (std::runtime_error: include/rcr/helpers.hpp:143-145, include/rcr/model.hpp:197-200), so code written
// against the reference keeps its error handling.  There is no CPU fallback behind this handle: without
// libsdm_hip.so's device the constructor throws.
#pragma once

#ifndef SDM_HIP_BACKEND_HPP_
#define SDM_HIP_BACKEND_HPP_

#include "sdm.h"

#include <memory>
#include <stdexcept>
#include <string>

namespace superviseddescent {
namespace hip {

inline void check(int rc, const char* what)
{
    if (rc < 0) throw std::runtime_error(std::string(what) + ": " + sdm_last_error());
}

// The device ordinal the header layer's handles are created on (per thread; default 0).  One process per GPU: a rank whose
// RCCL communicator lives on device `local_rank` calls set_device(local_rank) -- or passes it to set_data_parallel_rccl --
// before train()/test()/detect(), unless the launcher isolates each process with HIP_VISIBLE_DEVICES (then 0 is right).
inline int& device_ordinal()
{
    static thread_local int d = 0;
    return d;
}
inline void set_device(int device) { device_ordinal() = device; }
inline int device() { return device_ordinal(); }

class Handle {
public:
    explicit Handle(int device = hip::device()) : ctx_(sdm_create(device))
    {
        if (!ctx_) throw std::runtime_error(std::string("sdm_create: ") + sdm_last_error());
    }
    ~Handle() { if (ctx_) sdm_destroy(ctx_); }
    Handle(const Handle&) = delete;
    Handle& operator=(const Handle&) = delete;
    sdm_ctx* get() const { return ctx_; }

private:
    sdm_ctx* ctx_;
};

// Data-parallel training of the batched backend (one process per GPU, every process holds a shard of the rows): the
// per-level exchange of {A^T A, A^T b} is installed here, per thread, before SupervisedDescentOptimiser::train is called
// -- either a callback (any transport) or an RCCL communicator the library calls itself.  n_train_global = the row count
// over all ranks (the N of Regulariser's MatrixNorm, regressors.hpp:133-136).  The reference is single-process; with the
// default (nothing installed) the backend behaves exactly like it.
struct DataParallel {
    sdm_allreduce_fn fn = nullptr;
    void* user = nullptr;
    void* rccl_comm = nullptr;          // ncclComm_t
    void* rccl_allreduce = nullptr;     // &ncclAllReduce of the RCCL the application links, or nullptr: the library loads librccl
    int world_size = 1;
    long long n_train_global = 0;
    // sharded factorisation of the summed system (sdm.h: sdm_set_solve_sharding*): this rank's number, and either two
    // callbacks or -- with rccl_comm -- ncclBroadcast / ncclAllGather (addresses may be nullptr: looked up by name)
    int rank = -1;                      // < 0: replicated solve
    sdm_bcast_fn bcast = nullptr;
    sdm_allgather_fn allgather = nullptr;
    void* rccl_broadcast = nullptr;
    void* rccl_allgather = nullptr;
    // reduce-scatter form of the Gram exchange on top of the sharded factorisation (sdm.h: sdm_set_reduce_scatter*)
    sdm_reduce_scatter_fn reduce_scatter = nullptr;
    bool rccl_reduce_scatter = false;
    void* rccl_reduce_scatter_fn = nullptr;
};
inline DataParallel& data_parallel()
{
    static thread_local DataParallel dp;
    return dp;
}
inline void set_data_parallel(sdm_allreduce_fn fn, void* user, int world_size, long long n_train_global)
{
    DataParallel& dp = data_parallel();
    dp = DataParallel();
    dp.fn = fn; dp.user = user; dp.world_size = world_size; dp.n_train_global = n_train_global;
}
// (device >= 0: the ordinal the communicator was created on; the backend's handles must run on that device)
inline void set_data_parallel_rccl(void* nccl_comm, void* nccl_allreduce_fn, int world_size, long long n_train_global, int device = -1)
{
    if (device >= 0) set_device(device);
    DataParallel& dp = data_parallel();
    dp = DataParallel();
    dp.rccl_comm = nccl_comm; dp.rccl_allreduce = nccl_allreduce_fn; dp.world_size = world_size; dp.n_train_global = n_train_global;
}
// opt in to the sharded factorisation on top of either exchange (call after set_data_parallel / set_data_parallel_rccl)
inline void set_solve_sharding(int rank, sdm_bcast_fn bcast, sdm_allgather_fn allgather)
{
    DataParallel& dp = data_parallel();
    dp.rank = rank; dp.bcast = bcast; dp.allgather = allgather;
}
inline void set_solve_sharding_rccl(int rank, void* nccl_broadcast_fn = nullptr, void* nccl_allgather_fn = nullptr)
{
    DataParallel& dp = data_parallel();
    dp.rank = rank; dp.rccl_broadcast = nccl_broadcast_fn; dp.rccl_allgather = nccl_allgather_fn;
}
// with the sharded factorisation: ship every rank the sum of its own tile columns instead of all-reducing the whole matrix
inline void set_reduce_scatter(sdm_reduce_scatter_fn fn) { data_parallel().reduce_scatter = fn; }
inline void set_reduce_scatter_rccl(void* nccl_reduce_scatter_fn = nullptr)
{
    DataParallel& dp = data_parallel();
    dp.rccl_reduce_scatter = true; dp.rccl_reduce_scatter_fn = nccl_reduce_scatter_fn;
}
inline void clear_data_parallel() { data_parallel() = DataParallel(); }
// applied by the batched backend to the context it trains on
inline void install_data_parallel(sdm_ctx* c)
{
    const DataParallel& dp = data_parallel();
    if (dp.rccl_comm) check(sdm_set_allreduce_rccl(c, dp.rccl_comm, dp.rccl_allreduce, dp.world_size), "sdm_set_allreduce_rccl");
    else if (dp.fn) check(sdm_set_allreduce(c, dp.fn, dp.user, dp.world_size), "sdm_set_allreduce");
    if (dp.rank >= 0 && dp.bcast && dp.allgather)
        check(sdm_set_solve_sharding(c, dp.rank, dp.world_size, dp.bcast, dp.allgather, dp.user), "sdm_set_solve_sharding");
    else if (dp.rank >= 0 && dp.rccl_comm)
        check(sdm_set_solve_sharding_rccl(c, dp.rccl_comm, dp.rank, dp.world_size, dp.rccl_broadcast, dp.rccl_allgather),
              "sdm_set_solve_sharding_rccl");
    else
        check(sdm_set_solve_sharding(c, 0, 0, nullptr, nullptr, nullptr), "sdm_set_solve_sharding");
    if (dp.rank >= 0 && dp.rccl_comm && dp.rccl_reduce_scatter)
        check(sdm_set_reduce_scatter_rccl(c, 1, dp.rccl_reduce_scatter_fn), "sdm_set_reduce_scatter_rccl");
    else
        check(sdm_set_reduce_scatter(c, dp.rank >= 0 ? dp.reduce_scatter : nullptr, dp.user), "sdm_set_reduce_scatter");
}

// one lazily created handle per thread for the stand-alone solver calls
inline Handle& default_handle()
{
    static thread_local std::unique_ptr<Handle> h;
    if (!h) h.reset(new Handle(device()));
    return *h;
}

}  // namespace hip
}  // namespace superviseddescent

#endif
