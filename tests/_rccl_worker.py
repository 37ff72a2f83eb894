"""Worker of tests/test_gpu_distributed.py: launched by torch.distributed.run with one rank per visible GPU (nccl), or with
several ranks per GPU (SDM_TEST_BACKEND=gloo).
Trains a small cascade (a) without a collective and (b) through parallel.make_torch_allreduce on the nccl (= RCCL)
backend, the engine sharing torch's stream, and checks that (b) == (a) when WORLD_SIZE is 1, or that every rank
ends with identical regressors when it is larger.  Prints RCCL_WORKER_OK on success."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from superviseddescent_amd import (HogTransform, HoGParam, LinearRegressor, Regulariser,  # noqa: E402
                                   SupervisedDescentOptimiser, ibug, parallel, synth)


def main():
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ.get("LOCAL_RANK", "0"))
    # SDM_TEST_BACKEND=gloo: several REAL processes on however many GPUs there are (gloo moves device tensors through the host, so
    # ranks may share a GPU, which RCCL refuses) -- the one-GPU box's way to run world sizes > 1 through the product's collectives
    backend = os.environ.get("SDM_TEST_BACKEND", "nccl")
    if backend != "nccl":
        local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    else:
        dist.init_process_group(backend)
    ids = ibug.RCR22_IDS
    params = [HoGParam(1, 3, 12, 4, 0.9), HoGParam(1, 3, 9, 4, 0.6)]
    images, boxes, gt = synth.make_faces(48, seed=5)
    xs, x0, idx = synth.make_samples(boxes, gt, ids, n_perturb=3, seed=6)
    ra, rb = parallel.shard_range(xs.shape[0], rank, world)
    reg = lambda: Regulariser(Regulariser.RegularisationType.MatrixNorm, 1.5, False)
    stream = torch.cuda.current_stream().cuda_stream

    def train(allreduce, rows, shard=False, scatter=False):
        sdo = SupervisedDescentOptimiser([LinearRegressor(reg()) for _ in params], device=local, stream=stream)
        hog = HogTransform(images, params, ids, ibug.RIGHT_EYE_IDS, ibug.LEFT_EYE_IDS, idx[rows])
        calls, shard_calls = [], []
        fn, coll = None, None
        if allreduce is not None:
            def fn(ptr, count, s):
                calls.append(count)
                return allreduce(ptr, count, s)
        if shard:
            b, g = parallel.make_torch_solve_collectives(local)
            coll = (lambda *a: (shard_calls.append("b"), b(*a))[1], lambda *a: (shard_calls.append("g"), g(*a))[1])
        x = sdo.train(xs[rows], x0[rows], None, hog, allreduce=fn, world_size=world, n_train_global=xs.shape[0],
                      rank=rank if shard else None, solve_collectives=coll,
                      reduce_scatter=parallel.make_torch_reduce_scatter(local) if scatter else None)
        return [r.x.copy() for r in sdo.regressors], x, (shard_calls if shard else calls)

    R_dist, x_dist, calls = train(parallel.make_torch_allreduce(local), slice(ra, rb))
    assert len(calls) == len(params) and all(c > 0 for c in calls), calls      # one exchange per cascade level
    # the same with the factorisation sharded over the ranks (dist.broadcast / dist.all_gather_into_tensor on the engine's
    # stream): bit-identical regressors for any world size
    R_shard, x_shard, scalls = train(parallel.make_torch_allreduce(local), slice(ra, rb), shard=True)
    tiles = -(-R_dist[0].shape[0] // 128)
    assert scalls.count("b") == len(params) * tiles and scalls.count("g") == len(params) * (-(-tiles // 4) + (1 if world > 1 else 0)), len(scalls)
    for a, b in zip(R_shard, R_dist):
        assert np.array_equal(a, b), float(np.abs(a - b).max())
    assert np.array_equal(x_shard, x_dist)
    # ... and with the Gram exchange as a reduce-scatter of the owned tile columns (dist.reduce_scatter_tensor) + the small all-reduce
    R_rs, x_rs, _ = train(parallel.make_torch_allreduce(local), slice(ra, rb), shard=True, scatter=True)
    for a, b in zip(R_rs, R_dist):
        assert float(np.linalg.norm((a - b).astype(np.float64))) <= 2e-5 * float(np.linalg.norm(b.astype(np.float64)))
    assert float(np.linalg.norm((x_rs - x_dist).astype(np.float64))) <= 1e-5 * float(np.linalg.norm(x_dist.astype(np.float64)))
    if world == 1:
        R_solo, x_solo, _ = train(None, slice(0, xs.shape[0]))
        for a, b in zip(R_dist, R_solo):
            assert np.array_equal(a, b), float(np.abs(a - b).max())            # sum over one rank = identity
        assert np.array_equal(x_dist, x_solo)
    else:
        # against single-process training on ALL rows: the same system up to the summation order of the ranks' Gram matrices
        R_solo, x_solo, _ = train(None, slice(0, xs.shape[0]))
        dR = max(float(np.linalg.norm((a - b).astype(np.float64))) / float(np.linalg.norm(b.astype(np.float64))) for a, b in zip(R_dist, R_solo))
        dx = float(np.linalg.norm((x_dist - x_solo[ra:rb]).astype(np.float64))) / float(np.linalg.norm(x_solo[ra:rb].astype(np.float64)))
        if rank == 0:
            print("RCCL_WORKER_VS_SOLO dR=%.3g dx=%.3g" % (dR, dx))
        assert dR <= 5e-5 and dx <= 1e-6, (dR, dx)      # (measured 1.7-2.1e-5 and 2.3e-8 at 2, 3 and 4 ranks: the ranks' Gram matrices are summed in another order)
    for R in R_dist + R_shard + R_rs:                                          # every rank solved the same system
        t = torch.from_numpy(R).cuda()
        lo, hi = t.clone(), t.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN)
        dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        assert torch.equal(lo, hi)
    assert parallel.global_row_count(rb - ra) == xs.shape[0]
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        print("RCCL_WORKER_OK world=%d exchanges=%d floats=%d" % (world, len(calls), calls[0]))


if __name__ == "__main__":
    main()
