// sdm_capi_exchange.hip -- the exchange of the normal equations between the GPUs of a node (callbacks or library-issued RCCL) and the sharded factorisation's collectives (C-ABI of include/sdm.h; shared declarations: sdm_capi_internal.h)
#include "sdm_capi_internal.h"

namespace sdm_capi {
// the sharded factorisation's collectives as the solver calls them: RCCL on the handle's communicator, or the registered callbacks
int shard_bcast_thunk(void* self, float* buf, size_t count, int root, hipStream_t stream)
{
    sdm_ctx* c = (sdm_ctx*)self;
    if (c->shard_comm)   // ncclBroadcast(sendbuff, recvbuff, count, ncclFloat32 = 7, root, comm, stream), in place
        return c->rccl_bcast(buf, buf, count, 7, root, c->shard_comm, stream);
    return c->shard_bcast(buf, count, root, (void*)stream, c->shard_user);
}
int shard_allgather_thunk(void* self, const float* send, float* recv, size_t count, hipStream_t stream)
{
    sdm_ctx* c = (sdm_ctx*)self;
    if (c->shard_comm)   // ncclAllGather(sendbuff, recvbuff, sendcount, ncclFloat32 = 7, comm, stream)
        return c->rccl_allgather(send, recv, count, 7, c->shard_comm, stream);
    return c->shard_allgather(send, recv, count, (void*)stream, c->shard_user);
}
}  // namespace sdm_capi

extern "C" {

namespace {
void* find_rccl_symbol(const char* name)
{
    // the RCCL already mapped into the process (torch's, the application's) wins; otherwise ROCm's
    void* fn = dlsym(RTLD_DEFAULT, name);
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (int i = 0; !fn && i < 3; ++i) {
        void* hnd = dlopen(names[i], RTLD_NOW | RTLD_GLOBAL);
        if (hnd) fn = dlsym(hnd, name);
    }
    return fn;
}
}  // namespace

int sdm_set_allreduce(sdm_ctx* c, sdm_allreduce_fn fn, void* user, int world_size)
{
    if (!c || world_size < 1) return fail(SDM_ERR_INVALID, "bad all-reduce registration");
    c->allreduce = fn; c->allreduce_user = user; c->world_size = world_size;
    c->rccl_comm = nullptr; c->rccl_allreduce = nullptr;
    return SDM_OK;
}

int sdm_set_allreduce_rccl(sdm_ctx* c, void* nccl_comm, void* nccl_allreduce_fn, int world_size)
{
    if (!c || world_size < 1) return fail(SDM_ERR_INVALID, "bad all-reduce registration");
    if (!nccl_comm) { c->rccl_comm = nullptr; c->rccl_allreduce = nullptr; c->world_size = 1; return SDM_OK; }
    typedef int (*fn_t)(const void*, void*, size_t, int, int, void*, hipStream_t);
    fn_t fn = (fn_t)nccl_allreduce_fn;
    if (!fn) fn = (fn_t)find_rccl_symbol("ncclAllReduce");
    if (!fn) return fail(SDM_ERR_COMM, "ncclAllReduce not found: pass its address, or make librccl.so loadable");
    c->rccl_comm = nccl_comm; c->rccl_allreduce = fn; c->world_size = world_size;
    c->allreduce = nullptr; c->allreduce_user = nullptr;
    return SDM_OK;
}

int sdm_set_reduce_scatter(sdm_ctx* c, sdm_reduce_scatter_fn fn, void* user)
{
    if (!c) return fail(SDM_ERR_INVALID, "null handle");
    c->reduce_scatter = fn; c->reduce_scatter_user = user; c->rccl_reduce_scatter = nullptr;
    return SDM_OK;
}

int sdm_set_reduce_scatter_rccl(sdm_ctx* c, int enable, void* nccl_reduce_scatter_fn)
{
    if (!c) return fail(SDM_ERR_INVALID, "null handle");
    c->reduce_scatter = nullptr; c->reduce_scatter_user = nullptr; c->rccl_reduce_scatter = nullptr;
    if (!enable) return SDM_OK;
    if (!nccl_reduce_scatter_fn) nccl_reduce_scatter_fn = find_rccl_symbol("ncclReduceScatter");
    if (!nccl_reduce_scatter_fn) return fail(SDM_ERR_COMM, "ncclReduceScatter not found: pass its address, or make librccl.so loadable");
    c->rccl_reduce_scatter = (int (*)(const void*, void*, size_t, int, int, void*, hipStream_t))nccl_reduce_scatter_fn;
    return SDM_OK;
}


int sdm_set_solve_sharding(sdm_ctx* c, int rank, int world_size, sdm_bcast_fn bcast, sdm_allgather_fn allgather, void* user)
{
    if (!c) return fail(SDM_ERR_INVALID, "null handle");
    c->shard_comm = nullptr; c->rccl_bcast = nullptr; c->rccl_allgather = nullptr;
    if (!bcast && !allgather) { c->shard_world = 0; c->shard_bcast = nullptr; c->shard_allgather = nullptr; return SDM_OK; }
    if (!bcast || !allgather || world_size < 1 || rank < 0 || rank >= world_size)
        return fail(SDM_ERR_INVALID, "sdm_set_solve_sharding: both collectives and 0 <= rank < world_size are required");
    c->shard_rank = rank; c->shard_world = world_size; c->shard_bcast = bcast; c->shard_allgather = allgather; c->shard_user = user;
    return SDM_OK;
}

int sdm_set_solve_sharding_rccl(sdm_ctx* c, void* nccl_comm, int rank, int world_size, void* nccl_broadcast_fn, void* nccl_allgather_fn)
{
    if (!c) return fail(SDM_ERR_INVALID, "null handle");
    c->shard_bcast = nullptr; c->shard_allgather = nullptr; c->shard_user = nullptr;
    if (!nccl_comm) { c->shard_world = 0; c->shard_comm = nullptr; return SDM_OK; }
    if (world_size < 1 || rank < 0 || rank >= world_size) return fail(SDM_ERR_INVALID, "sdm_set_solve_sharding_rccl: 0 <= rank < world_size required");
    if (!nccl_broadcast_fn) nccl_broadcast_fn = find_rccl_symbol("ncclBroadcast");
    if (!nccl_allgather_fn) nccl_allgather_fn = find_rccl_symbol("ncclAllGather");
    if (!nccl_broadcast_fn || !nccl_allgather_fn)
        return fail(SDM_ERR_COMM, "ncclBroadcast / ncclAllGather not found: pass their addresses, or make librccl.so loadable");
    c->rccl_bcast = (int (*)(const void*, void*, size_t, int, int, void*, hipStream_t))nccl_broadcast_fn;
    c->rccl_allgather = (int (*)(const void*, void*, size_t, int, void*, hipStream_t))nccl_allgather_fn;
    c->shard_comm = nccl_comm; c->shard_rank = rank; c->shard_world = world_size;
    return SDM_OK;
}


int sdm_allreduce_gram_rhs(sdm_ctx* c)
{
    if (!c || c->g_level < 0) return fail(SDM_ERR_INVALID, "no Gram matrix to reduce");
    if (!c->allreduce && !c->rccl_comm) return SDM_OK;
    Timer t(c, SDM_T_ALLREDUCE);
    // only the tiles the solve reads travel: upper Gram tiles + RHS tile columns, packed back to back
    HIP_TRY(hipSetDevice(c->device));
    const int F = level_F(c, c->g_level);
    // Reduce-scatter instead, when the factorisation is sharded over the same ranks: a rank's share of the factorisation reads its
    // own tile columns only, so each rank needs the SUM of its columns, not of the whole matrix -- half the bytes on the ring.  What
    // every rank needs of the others' columns is small and follows in ONE all-reduce of F + 1 floats: the summed diagonal (the
    // float16 updates' scale is taken from its largest entry) and the ranks' shares of ||G||_F^2 (MatrixNorm).
    const bool sharded = c->shard_world >= 1 && (c->shard_comm || c->shard_bcast) && c->shard_world == c->world_size;
    const bool can_rs = c->reduce_scatter || (c->rccl_reduce_scatter && c->rccl_comm);
    // (the column-pivoted QR is replicated: its levels take the all-reduce of the whole matrix, as sdm_gram_rhs assumed -- ADVICE r04)
    if (sharded && can_rs && c->solver_kind == SDM_SOLVER_CHOLESKY) {
        const int W = c->shard_world, me = c->shard_rank;
        int rc;
        if ((rc = c->gsmall.ensure((size_t)F + 1)) || (rc = c->fro.ensure((size_t)F + 1))) return rc;
        int rcn = 0;
        auto scatter = [&](float* send, float* recv, size_t count, hipStream_t st) {
            if (c->rccl_reduce_scatter)   // ncclReduceScatter(sendbuff, recvbuff, recvcount, ncclFloat32 = 7, ncclSum = 0, comm, stream)
                return c->rccl_reduce_scatter(send, recv, count, 7, 0, c->rccl_comm, st);
            return c->reduce_scatter(send, recv, count, (void*)st, c->reduce_scatter_user);
        };
        if (c->gram_blocks > 1 && c->solve_aux.stream) {
            // The Gram matrix was multiplied in ranges of tile columns with an event behind each (sdm_gram_rhs): the second queue ships
            // range b -- pack, reduce-scatter of the ranks' chunks of that range, unpack of the own chunk -- as soon as ITS products are
            // done, while the kernel is still busy with the ranges behind it.  The per-element sums are those of the one-piece exchange.
            const int nb = c->gram_blocks;
            size_t total = 0, chunk_b[sdm_ctx::XBLOCKS_MAX];
            for (int b = 0; b < nb; ++b) {
                chunk_b[b] = sdm_owned_chunk_tiles(F, c->rhs_tiles, W, c->gram_block_c[b], c->gram_block_c[b + 1]) * 128 * 128;
                total += (size_t)(W + 1) * chunk_b[b];
            }
            if ((rc = c->gpack.ensure(total))) return rc;
            hipStream_t xs = c->solve_aux.stream;
            float* base = c->gpack.p;
            for (int b = 0; b < nb && rcn == 0; ++b) {
                float* send = base;
                float* recv = base + (size_t)W * chunk_b[b];
                base += (size_t)(W + 1) * chunk_b[b];
                HIP_TRY(hipStreamWaitEvent(xs, c->gram_ev[b], 0));
                if (chunk_b[b] == 0) continue;
                sdm_launch_tiles_pack_owned(c->G.p, c->g_ncols, F, c->rhs_tiles, W, me, send, 0, xs, c->gram_block_c[b], c->gram_block_c[b + 1]);
                rcn = scatter(send, recv, chunk_b[b], xs);
                if (rcn == 0) sdm_launch_tiles_pack_owned(c->G.p, c->g_ncols, F, c->rhs_tiles, W, me, recv, 1, xs, c->gram_block_c[b], c->gram_block_c[b + 1]);
            }
            HIP_TRY(hipEventRecord(c->gram_xdone, xs));
            HIP_TRY(hipStreamWaitEvent(c->stream, c->gram_xdone, 0));      // (also on the error path: the caller's stream owns G again)
            if (rcn != 0) return fail(SDM_ERR_COMM, "reduce-scatter failed with status " + std::to_string(rcn));
            HIP_TRY(hipGetLastError());
        } else {
        const size_t chunk = sdm_owned_chunk_tiles(F, c->rhs_tiles, W) * 128 * 128;
        if ((rc = c->gpack.ensure((size_t)(W + 1) * chunk))) return rc;
        float* send = c->gpack.p;
        float* recv = c->gpack.p + (size_t)W * chunk;
        sdm_launch_tiles_pack_owned(c->G.p, c->g_ncols, F, c->rhs_tiles, W, me, send, 0, c->stream);
        HIP_TRY(hipGetLastError());
        rcn = scatter(send, recv, chunk, c->stream);
        if (rcn != 0) return fail(SDM_ERR_COMM, "reduce-scatter failed with status " + std::to_string(rcn));
        sdm_launch_tiles_pack_owned(c->G.p, c->g_ncols, F, c->rhs_tiles, W, me, recv, 1, c->stream);
        }
        // the small exchange: [diagonal of the owned columns, 0 elsewhere | this rank's share of ||G||_F^2]
        sdm_launch_diag_owned(c->G.p, c->g_ncols, F, W, me, c->gsmall.p, 0, c->stream);
        sdm_launch_fro2_upper(c->G.p, c->g_ncols, F, c->fro.p, c->stream, me, W);
        sdm_launch_small_exchange_pack(c->fro.p + F, c->gsmall.p + F, 0, nullptr, c->stream);
        HIP_TRY(hipGetLastError());
        if (c->rccl_comm) rcn = c->rccl_allreduce(c->gsmall.p, c->gsmall.p, (size_t)F + 1, 7, 0, c->rccl_comm, c->stream);
        else rcn = c->allreduce(c->gsmall.p, (size_t)F + 1, (void*)c->stream, c->allreduce_user);
        if (rcn != 0) return fail(SDM_ERR_COMM, "all-reduce of the diagonal failed with status " + std::to_string(rcn));
        sdm_launch_diag_owned(c->G.p, c->g_ncols, F, W, me, c->gsmall.p, 1, c->stream);
        sdm_launch_small_exchange_pack(nullptr, c->gsmall.p + F, 1, c->fro.p + F, c->stream);
        HIP_TRY(hipGetLastError());
        c->g_scattered = true;
        return SDM_OK;
    }
    const size_t count = sdm_packed_tiles_count(F, c->rhs_tiles);
    int rc = c->gpack.ensure(count);
    if (rc) return rc;
    sdm_launch_tiles_pack(c->G.p, c->g_ncols, F, c->rhs_tiles, c->gpack.p, 0, c->stream);
    HIP_TRY(hipGetLastError());
    if (c->rccl_comm) {
        // ncclAllReduce(sendbuff, recvbuff, count, ncclFloat32 = 7, ncclSum = 0, comm, stream): in place, on the engine's stream,
        // i.e. ordered behind the pack kernel and before the unpack without any host synchronisation
        const int rcn = c->rccl_allreduce(c->gpack.p, c->gpack.p, count, 7, 0, c->rccl_comm, c->stream);
        if (rcn != 0) return fail(SDM_ERR_COMM, "ncclAllReduce failed with status " + std::to_string(rcn));
    } else if (c->allreduce(c->gpack.p, count, (void*)c->stream, c->allreduce_user) != 0)
        return fail(SDM_ERR_COMM, "all-reduce callback reported failure");
    sdm_launch_tiles_pack(c->G.p, c->g_ncols, F, c->rhs_tiles, c->gpack.p, 1, c->stream);
    HIP_TRY(hipGetLastError());
    return SDM_OK;
}


}  // extern "C"
