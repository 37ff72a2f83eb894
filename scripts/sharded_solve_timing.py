"""Per-rank compute time of the sharded factorisation, measured on ONE GPU (no multi-GPU box in reach): a single context
runs rank r's share of the launches -- its owned tile columns of every row update, panel solve and trailing update, the potrf of
EVERY diagonal tile (SDM_SOLVE_SHARD_EMULATE=1: a rank waits for the owner's potrf, so the time belongs to its chain), the pack / unpack kernels of both exchanges, the replicated back substitution -- with collectives
that return immediately.  What a real run adds is the time of the collectives themselves; their count and bytes are printed
(one broadcast per 128-column step, one all-gather per group of 4 steps).  The data the no-op collectives leave behind is
never arrives, so the stand-in collectives write zero panel tiles and a scaled identity as the owner's diagonal factor, and the
system is made diagonally dominant (Manual lambda = 1e12): every potrf succeeds and runs its full length.

    python scripts/sharded_solve_timing.py [rcr22 rcr68]      -> gpurun_out/sharded_solve_timing.json
"""
import json
import os
import sys

import numpy as np

os.environ["SDM_SOLVE_SHARD_EMULATE"] = "1"      # also run the potrf of the tiles other ranks own: the chain every rank waits for
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from superviseddescent_amd import Context, HoGParam, ibug, synth  # noqa: E402

CONFIGS = {"rcr22": (ibug.RCR22_IDS, 8801), "rcr68": (ibug.IBUG68_IDS, 27201)}


class _Span:
    def __init__(self, ptr, count):
        self.__cuda_array_interface__ = {"shape": (int(count),), "typestr": "<f4", "data": (int(ptr), False), "version": 3,
                                         "strides": None}


def main():
    import torch
    which = sys.argv[1:] or ["rcr22", "rcr68"]
    out = {}
    dev = torch.device("cuda", 0)
    diag_tile = (torch.eye(128, device=dev) * 1e6).reshape(-1).contiguous()
    current = {"rank": 0}
    for name in which:
        ids, F = CONFIGS[name]
        re, le = ibug.eye_indices(ids)
        images, boxes, gt = synth.make_faces(128, seed=1)
        x_star, x0, idx = synth.make_samples(boxes, gt, ids, n_perturb=3, seed=2)          # 512 rows: the Gram content is irrelevant
        ctx = Context(0)
        ctx.set_model_geometry(len(ids), re, le, [HoGParam(*ibug.SHIPPED_HOG_PARAMS[0])])
        ctx.upload_images(images)
        ctx.set_sample_image_index(idx)
        ctx.set_x(x0)
        ctx.set_targets(x_star)
        ctx.enable_timing(True)
        Tf = -(-F // 128)
        T = Tf + (-(-(2 * len(ids)) // 16) * 16 + 127) // 128

        def solve_ms(reps=3):
            best = 1e9
            for _ in range(reps):
                ctx.set_x(x0)
                ctx.hog_features(0)
                ctx.gram_rhs(0)
                ctx.get_timing(reset=True)
                ctx.solve(0, 0, 1e12, True, n_train_global=0, fetch=False)
                best = min(best, ctx.get_timing(reset=True)["factor_solve"][0])
            return best
        res = {"features": F, "factor_tiles": Tf, "tile_columns": T, "replicated_ms": solve_ms()}
        print(name, "replicated: %.2f ms" % res["replicated_ms"], flush=True)
        for world in (2, 4, 8):
            traffic = {"bcast": 0, "bcast_bytes": 0, "allgather": 0, "allgather_bytes_recv": 0}

            def bcast(ptr, count, root, stream, world=world):
                traffic["bcast"] += 1
                traffic["bcast_bytes"] += 4 * count
                if root != current["rank"]:      # stand-in for the owner's data: zero panel tiles, a well-scaled diagonal factor
                    with torch.cuda.stream(torch.cuda.ExternalStream(int(stream), device=dev)):
                        v = torch.as_tensor(_Span(ptr, count), device=dev)
                        v[:count - 128 * 128].zero_()
                        v[count - 128 * 128:].copy_(diag_tile)
                return 0

            def allgather(send, recv, count, stream, world=world):
                traffic["allgather"] += 1
                traffic["allgather_bytes_recv"] += 4 * count * (world - 1)
                with torch.cuda.stream(torch.cuda.ExternalStream(int(stream), device=dev)):
                    torch.as_tensor(_Span(recv, count * world), device=dev).zero_()
                return 0
            per_rank = {}
            for rank in sorted({0, 1, world - 1}):
                current["rank"] = rank
                ctx.set_solve_sharding(rank, world, bcast, allgather)
                for k in traffic:
                    traffic[k] = 0
                ms = solve_ms()
                per_rank[rank] = ms
                calls = 3            # solve_ms repeats
                print(name, "world %d rank %d: %.2f ms  (%d bcasts / %.1f MB, %d all-gathers / %.1f MB received per solve)" % (
                    world, rank, ms, traffic["bcast"] // calls, traffic["bcast_bytes"] / calls / 1e6, traffic["allgather"] // calls,
                    traffic["allgather_bytes_recv"] / calls / 1e6), flush=True)
            res["world_%d" % world] = {"per_rank_compute_ms": per_rank, "bcasts": traffic["bcast"] // 3,
                                       "bcast_mb": traffic["bcast_bytes"] / 3 / 1e6, "allgathers": traffic["allgather"] // 3,
                                       "allgather_mb_received": traffic["allgather_bytes_recv"] / 3 / 1e6}
            ctx.set_solve_sharding(0, 0, None, None)
        out[name] = res
        ctx.close()
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", "sharded_solve_timing.json"), "w") as fh:
            json.dump(out, fh, indent=1)


if __name__ == "__main__":
    main()
