#!/usr/bin/env python
"""bench.py -- RCR-22 detect throughput on MI355X (BASELINE.json metric "faces/sec RCR-22 detect
(batch 4096)").

One step = one full cascade (4 levels of batched HOG extraction + regressor apply + update) over a batch
of 4096 synthetic 256x256 faces per GPU.  Images, the initial landmark rows and the model are resident
in HBM before the timed region.  With N GPUs every rank detects its own 4096 faces (independent shards,
no collective on the data path -> weak scaling); `value` = faces of all ranks / max-over-ranks time.

    python bench.py --gpus 1 --steps 200 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        --master-port 29500 bench.py --gpus 8 --steps 200 --warmup 5

Rank 0 prints ONE JSON line (contract in the task description) including
  "roofline"     -- HOG kernel: algorithmic HBM bytes / HIP-event-timed kernel duration vs 8 TB/s
  "cpu_baseline" -- the CPU oracle (the reference's algorithm restated, oracle/; its HOG back-end is the reference's own
                    hog.c when oracle/_ref is present) timed on this box's cores on a bounded sample of the same workload
  "parity"       -- GPU vs oracle on that sample: landmarks (relative L2), faces whose integer patch decisions differ at any
                    level, the largest per-face error, and the same in SDM_HOG_EXACT_ORDER mode
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_F32_PEAK_TF = 157.3   # f32-input MFMA dense peak
RCR68_DEADLINE_S = float(os.environ.get("SDM_BENCH_RCR68_DEADLINE_S", "420"))      # several GPUs: watchdog of the RCR-68 legs (they take ~10 s on one GPU)
MFMA_F16_PEAK_TF = 2500.0  # f16 / bf16-input MFMA dense peak (the Gram launch: four float16 piece products per f32 product)


def gram_report(exec_flops, useful_flops, ms):
    """The Gram launch forms every f32 product from three float16 piece products on the 16-bit matrix cores (csrc/sdm_gram_bf16.hip):
    `achieved` counts f32-equivalent flops (upper 128 x 128 tiles incl. padding + right-hand-side tile columns), `peak` is the f16
    matrix-core peak / 3; the split pre-pass is inside the timed stage."""
    if ms <= 0:
        return None
    tf = exec_flops / (ms * 1e-3) / 1e12
    return {"kernel": "split_planes_f16_kernel+syrk_tn_split_w4_kernel", "bound": "mfma", "unit": "TFLOP/s (f32-equivalent)",
            "peak": MFMA_F16_PEAK_TF / 3.0, "achieved": tf, "frac": tf / (MFMA_F16_PEAK_TF / 3.0),
            "useful_tflops": useful_flops / (ms * 1e-3) / 1e12, "times_f32_mfma_peak": tf / MFMA_F32_PEAK_TF,
            "stage_ms": ms,
            "matrix_pipe": "profiles/r06_gram_pmc.txt: SQ_VALU_MFMA_BUSY_CYCLES / (1 024 SIMDs x GRBM_GUI_ACTIVE / 8) = 0.93 of the clocks, which "
                           "the power limit holds at ~1.35 GHz under this load (2.4 GHz nominal = the clock `peak` is quoted at); "
                           "profiles/r06_gram_power_check.txt: the same launch on rows of zeros takes 25 % less time than on the feature rows; "
                           "profiles/r06_mfma_power.txt: back-to-back matrix instructions on register operands sustain 2.49 PFLOP/s on zeros, 1.55 on random float16",
            "note": "every f32 operand = two float16 pieces (x 2^12), three piece products per product (low x low is below float32's "
                    "rounding), float32 accumulation; measured against a float64 product: 1.1e-7 ... 2.9e-7 relative (the f32 "
                    "matrix-core kernel it replaces: 2.4e-7 ... 3.5e-7)"}


def apply_report(flops_per_launch, feature_bytes_per_launch, ms):
    """The UNFUSED regressor apply (LDS-staged GEMM over the feature matrix, csrc/sdm_apply.hip; training's update step and
    SDM_DETECT_UNFUSED=1): every f32 product = three float16 piece products on the 16-bit matrix cores.  One fraction per pipe the
    kernel runs on: `mfma_f16` = executed piece flops / f16 peak, `hbm` = the feature matrix read once."""
    if ms <= 0:
        return None
    tf = flops_per_launch / (ms * 1e-3) / 1e12
    gbs = feature_bytes_per_launch / (ms * 1e-3) / 1e9
    return {"kernel": "apply_tiled_f16_kernel+apply_reduce_kernel", "bound": "hbm", "unit": "GB/s", "achieved": gbs, "peak": HBM_PEAK_GBS,
            "frac": gbs / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": feature_bytes_per_launch,
            "mfma_f16": {"achieved": 3.0 * tf, "peak": MFMA_F16_PEAK_TF, "unit": "TFLOP/s (float16 piece products executed)",
                         "frac": 3.0 * tf / MFMA_F16_PEAK_TF},
            "f32_equivalent_tflops": tf, "avg_launch_ms": ms,
            "arithmetic_intensity_flop_per_byte": flops_per_launch / feature_bytes_per_launch if feature_bytes_per_launch else None,
            "note": "[N x F] . [F x M] with M = 2L: 2 N F M flop over 4 N F feature bytes = M / 2 flop per byte (22 at RCR-22, 68 at "
                    "RCR-68; SURVEY 8d counts 21.6 with the outputs).  The f16-piece matrix pipe's ridge is 2 500 TF x 3 / 8 TB/s = "
                    "940 executed flop per byte: the product is HBM-bound by construction (or latency-bound where the launch is short), "
                    "so north_star's '>= 50 % MFMA utilisation for the regressor GEMM' cannot be met by the APPLY product at any kernel "
                    "quality -- it is met (or not) by the Gram product, the GEMM the training metric spends its time in (`gram`); "
                    "the apply's own bound is `hbm` above"}


def fused_apply_report(n_faces, L, P, M, cut_frac, ms):
    """The fused descriptor + apply launch of detect (csrc/sdm_desc.hip: desc_kernel<FUSED> + apply_reduce_kernel): reads the raw
    cells, normalises them, multiplies [faces x P] by the landmark's [P x 2L] regressor slice on the 16-bit matrix cores (three
    float16 piece products per product) and writes partial[L][faces][2L]; the N x F feature matrix is never written.
    `hbm`: cells read (+ the second part of patches cut by a pass boundary) + partial written and read back by the reduction.
    `mfma_f16`: executed piece flops (K padded to 32, 2L to 16) against the f16 peak -- the launch is bound by neither: its
    workgroups are latency chains (cells load -> ~270 vector instructions per patch pair -> barrier -> fragment loads -> 39 matrix
    instructions), three resident per CU."""
    if ms <= 0:
        return None
    Mp, KP = (M + 15) // 16 * 16, (P + 31) // 32 * 32
    cells = n_faces * L * 2 * 8 * 25 * 4.0 / 2 * (1.0 + cut_frac)      # 800 B per part
    partial = 2.0 * L * n_faces * Mp * 4.0
    gbs = (cells + partial) / (ms * 1e-3) / 1e9
    tf_exec = 3.0 * 2.0 * n_faces * L * KP * Mp / (ms * 1e-3) / 1e12
    return {"kernel": "desc_kernel<FUSED>+apply_reduce_kernel", "bound": "latency", "avg_stage_ms": ms,
            "hbm": {"achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "bytes_per_launch": cells + partial},
            "mfma_f16": {"achieved": tf_exec, "peak": MFMA_F16_PEAK_TF, "unit": "TFLOP/s (float16 piece products executed)",
                         "frac": tf_exec / MFMA_F16_PEAK_TF},
            "f32_equivalent_tflops": 2.0 * n_faces * (L * P + 1) * M / (ms * 1e-3) / 1e12,
            "feature_matrix_bytes_not_written": n_faces * (L * P + 1) * 4.0,
            "note": "arithmetic intensity M / 2 = 22 flop per feature byte (SURVEY 8d: 21.6) against a ridge of ~940 for the f16-piece "
                    "pipe: HBM- or, as here, latency-bound by construction; the north star's 50 % MFMA target is unreachable for this "
                    "product and is carried by the Gram product (`train.gram`, `rcr68_train.gram`)"}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=4096, help="faces per GPU")
    ap.add_argument("--train-rows", type=int, default=100000,
                    help="training rows = faces of the train-metric leg (images x 10 initialisations), sharded over the GPUs")
    ap.add_argument("--cpu-sample", type=int, default=0,
                    help="faces of the batch timed on the CPU oracle (0 = about 20 core-seconds of work)")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--rcr68-shard", type=int, default=8192,
                    help="faces per GPU of the RCR-68 detect leg (BASELINE config 4: 65 536 faces over 8 GPUs); 0 skips the RCR-68 legs")
    return ap.parse_args()


def relaunch_under_torchrun(args):
    """`python bench.py --gpus N` started WITHOUT a launcher (no RANK / WORLD_SIZE in the environment) replaces itself by
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py <same arguments>`: one rank per GPU, as the driver
    launches the N > 1 lines.  Without this the process would run ONE rank and print n_gpus = 1."""
    import socket
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"), SDM_BENCH_RELAUNCHED="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    os.execvpe(sys.executable, cmd, env)


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        if os.environ.get("SDM_BENCH_RELAUNCHED") == "1":
            raise SystemExit("bench.py: relaunched under torch.distributed.run but no WORLD_SIZE arrived")
        relaunch_under_torchrun(args)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    # ---- training rows of THIS rank (metric "train sec/cascade 100k faces"): images x (1 + 9 perturbed boxes) as in
    # rcr-train.cpp:421-431.  Generated first, in forked numpy workers, before torch / HIP are initialised.
    from superviseddescent_amd import ibug, parallel, synth
    t0 = time.time()
    ids = ibug.RCR22_IDS
    rows_per_image = 10
    n_img_total = max(args.train_rows // rows_per_image, 8 * world)
    ia, ib = parallel.shard_range(n_img_total, rank, world)
    workers = max(1, min(16, (os.cpu_count() or 1) // (2 * world)))
    timg, tbox, tgt = synth.make_faces(ib - ia, seed=synth.SEED + 1000 + rank, chunk=32,
                                       workers=workers if ib - ia >= 1024 else 0)
    txs, tx0, tidx = synth.make_samples(tbox, tgt, ids, n_perturb=rows_per_image - 1, seed=synth.SEED + 2000 + rank)
    n_train_global = n_img_total * rows_per_image
    # RCR-68 rows on the SAME images (BASELINE config 5: RCR-68 train, 100k faces): 68 landmarks, same boxes and perturbations
    ids68 = ibug.IBUG68_IDS
    if args.rcr68_shard > 0:
        txs68, tx068, tidx68 = synth.make_samples(tbox, tgt, ids68, n_perturb=rows_per_image - 1, seed=synth.SEED + 2000 + rank)
    # detect faces of this rank: the headline batch = the first args.batch of them, the RCR-68 shard = all of them
    n_detect = max(args.batch, args.rcr68_shard)
    images, boxes, gt = synth.make_faces(n_detect, seed=synth.SEED + 17 * rank, chunk=32,
                                         workers=workers if n_detect >= 1024 else 0)
    datagen_s = time.time() - t0

    import torch
    import torch.distributed as dist
    from superviseddescent_amd import (Context, HoGParam, HogTransform, LinearRegressor, Regulariser,
                                       SupervisedDescentOptimiser)

    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: there is no CPU fallback for the measured path")
    # SDM_BENCH_BACKEND=gloo (tests): the several-GPU path run by several real processes on however many GPUs there are -- gloo moves
    # device tensors through the host, so two ranks may share one GPU, which RCCL refuses.  The ranks then share devices round-robin.
    backend = os.environ.get("SDM_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    # launched by torch.distributed.run (RANK/WORLD_SIZE/MASTER_* in the env): the collective path is taken even
    # at WORLD_SIZE=1, so that one GPU exercises exactly the code N GPUs run
    use_dist = world > 1 or ("RANK" in os.environ and "MASTER_PORT" in os.environ)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    re, le = ibug.eye_indices(ids)
    params = [HoGParam(*p) for p in ibug.SHIPPED_HOG_PARAMS]   # apps/rcr/rcr-train.cpp:447
    n_levels = len(params)
    L, M = len(ids), 2 * len(ids)
    stream = torch.cuda.current_stream().cuda_stream

    # ---- model + secondary metric "train sec/cascade": an RCR-22 cascade trained here on the synthetic rows generated
    # above (the shipped .bin models are not in the reference checkout).  With N GPUs the ROWS are sharded (strong
    # scaling) and {A^T A, A^T b} are summed with one RCCL all-reduce per cascade level, after which every rank solves
    # the identical system ---------------------------------------------------------------------------------------
    t0 = time.time()
    reg = lambda: Regulariser(Regulariser.RegularisationType.MatrixNorm, 1.5, False)   # rcr-train.cpp:440-443
    sdo = SupervisedDescentOptimiser([LinearRegressor(reg()) for _ in params], device=local_rank, stream=stream)
    hog = HogTransform(timg, params, ids, ibug.RIGHT_EYE_IDS, ibug.LEFT_EYE_IDS, tidx, images_resident=True)   # (uploaded by the first pass)
    # The exchange.  Backend nccl (= RCCL): the LIBRARY issues the collectives itself -- ncclAllReduce / ncclReduceScatter /
    # ncclBroadcast / ncclAllGather on its own HIP streams, through a communicator of this process's own (parallel.RcclCommunicator:
    # ncclCommInitRank from a unique id carried by torch.distributed; sdm_set_allreduce_rccl, sdm_set_reduce_scatter_rccl,
    # sdm_set_solve_sharding_rccl).  No Python runs between two kernels of a training level: the 1 reduce-scatter (x 4 ranges) +
    # 213 broadcasts + 55 all-gathers of an RCR-68 level are queued by the same host loop that queues the kernels (VERDICT r04 item
    # 6).  The torch.distributed callbacks (ctypes -> Python -> torch) remain for backends without RCCL -- the gloo runs of the
    # tests -- and as SDM_BENCH_COLLECTIVES=torch for an A/B.
    native = use_dist and backend == "nccl" and os.environ.get("SDM_BENCH_COLLECTIVES", "rccl") != "torch"
    # The communicator is created at its FIRST USE -- with several GPUs that is behind the headline measurement, under the watchdog
    # below: an ncclCommInitRank that hangs or fails on links this code has never seen must not take the headline with it.
    class _LazyRccl:
        def __init__(self):
            self.comm = None

        def get(self):
            if self.comm is None:
                self.comm = parallel.RcclCommunicator(rank, world)
            return self.comm

        def destroy(self):
            if self.comm is not None:
                self.comm.destroy()
    lazy_rccl = _LazyRccl() if native else None
    allreduce = parallel.make_torch_allreduce(local_rank) if (use_dist and not native) else None
    # SDM_BENCH_SHARD_SOLVE=1: the summed system is factored by all ranks together (tile-column ownership).  Off by
    # default: at the bench's F = 8 801 the factorisation is bound by its chain of 69 single-workgroup panel steps, which
    # sharding does not shorten (profiles/r02_sharded_solve_timing.json); it pays at F = 27 201 (RCR-68)
    shard_solve = use_dist and os.environ.get("SDM_BENCH_SHARD_SOLVE", "0") == "1"
    solve_collectives = parallel.make_torch_solve_collectives(local_rank) if (shard_solve and not native) else None
    # with a sharded solve the exchange is a reduce-scatter of the owned tile columns + a small all-reduce (half the ring traffic)
    reduce_scatter = parallel.make_torch_reduce_scatter(local_rank) if (use_dist and not native) else None
    via = ("RCCL called by the library on its own streams (sdm_set_*_rccl, own ncclCommInitRank)" if native else
           "torch.distributed %s through the sdm_set_* callbacks" % backend)
    def train_rcr22(collective, sdo=sdo, hog=hog):
        """Two passes of sdo.train (the second one timed).  collective=False: every rank on its own rows, no exchange -- the
        model the detect legs run with several GPUs (see defer below)."""
        nlsr_, wall_, timing_ = [], [], None
        for rep in range(2):    # the second pass is the measured one (buffers allocated, code loaded)
            sdo.ctx.enable_timing(True)
            sdo.ctx.get_timing(reset=True)
            if use_dist and collective:
                dist.barrier()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            if collective and native:
                sdo.train(txs, tx0, None, hog, world_size=world, n_train_global=n_train_global, rccl=lazy_rccl.get(), rccl_shard_solve=shard_solve,
                          on_training_epoch_callback=(lambda cur: nlsr_.append(float(np.linalg.norm(cur - txs) / np.linalg.norm(txs))))
                          if rep == 0 else None)
            elif collective:
                sdo.train(txs, tx0, None, hog, allreduce=allreduce, world_size=world, n_train_global=n_train_global,
                          rank=rank if shard_solve else None, solve_collectives=solve_collectives,
                          reduce_scatter=reduce_scatter if shard_solve else None,
                          on_training_epoch_callback=(lambda cur: nlsr_.append(float(np.linalg.norm(cur - txs) / np.linalg.norm(txs))))
                          if rep == 0 else None)
            else:
                sdo.train(txs, tx0, None, hog,
                          on_training_epoch_callback=(lambda cur: nlsr_.append(float(np.linalg.norm(cur - txs) / np.linalg.norm(txs))))
                          if rep == 0 else None)
            if use_dist and collective:
                dist.barrier()
            torch.cuda.synchronize()
            wall_.append(time.perf_counter() - t1)
            timing_ = sdo.ctx.get_timing(reset=True)
        if use_dist and collective:
            tt = torch.tensor([wall_[-1]], dtype=torch.float64, device="cuda")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            wall_[-1] = float(tt.item())
        sdo.ctx.set_allreduce(None, 1)
        sdo.ctx.set_solve_sharding(0, 0, None, None)
        sdo.ctx.set_reduce_scatter(None)
        return list(nlsr_), wall_, timing_

    # Several GPUs: everything that needs a collective -- the RCR-22 training exchange, the RCR-68 legs -- runs BEHIND the headline
    # measurement and under a watchdog (below): the exchanges have only ever run with ranks as threads on one GPU
    # (tests/test_gpu_sharded_solve.py, test_gpu_exchange.py), and a collective that hangs on real links must not take the headline
    # of the scaling run with it.  The detect legs then run with a model every rank trained on its own rows (its coefficients do not
    # enter the timing; the parity block compares against the oracle with the same coefficients).
    defer68 = world > 1 or os.environ.get("SDM_BENCH_DEFER_RCR68", "0") == "1"      # (the switch: the several-GPU order on one GPU, for the test)
    nlsr, train_wall, train_timing = train_rcr22(collective=not defer68)
    train_s = time.time() - t0
    regressors = [r.x for r in sdo.regressors]
    ctx = sdo.ctx

    # ---- BASELINE config 5: RCR-68 (iBUG-68, F = 27 201, M = 136) trained on the same 100k rows, sharded over the ranks.  The
    # summed system is large enough for the sharded factorisation to pay, so with N > 1 GPUs it is on by default
    # (SDM_BENCH_SHARD_SOLVE=0 keeps the replicated solve) -------------------------------------------------------------------
    def train_rcr68():
        L68, M68 = len(ids68), 2 * len(ids68)
        sdo68 = SupervisedDescentOptimiser([LinearRegressor(reg()) for _ in params], device=local_rank, stream=stream)
        hog68 = HogTransform(timg, params, ids68, ibug.RIGHT_EYE_IDS, ibug.LEFT_EYE_IDS, tidx68, images_resident=True)
        shard68 = use_dist and world > 1 and os.environ.get("SDM_BENCH_SHARD_SOLVE", "1") == "1"
        coll68 = parallel.make_torch_solve_collectives(local_rank) if (shard68 and not native) else None
        nlsr68, wall68 = [], []
        for rep in range(2):
            sdo68.ctx.enable_timing(True)
            sdo68.ctx.get_timing(reset=True)
            if use_dist:
                dist.barrier()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            cb68 = (lambda cur: nlsr68.append(float(np.linalg.norm(cur - txs68) / np.linalg.norm(txs68)))) if rep == 0 else None
            if native:
                sdo68.train(txs68, tx068, None, hog68, world_size=world, n_train_global=n_train_global, rccl=lazy_rccl.get(), rccl_shard_solve=shard68,
                            on_training_epoch_callback=cb68)
            else:
                sdo68.train(txs68, tx068, None, hog68, allreduce=allreduce, world_size=world, n_train_global=n_train_global,
                            rank=rank if shard68 else None, solve_collectives=coll68, reduce_scatter=reduce_scatter if shard68 else None,
                            on_training_epoch_callback=cb68)
            if use_dist:
                dist.barrier()
            torch.cuda.synchronize()
            wall68.append(time.perf_counter() - t1)
            timing68 = sdo68.ctx.get_timing(reset=True)
        if use_dist:
            tt = torch.tensor([wall68[-1]], dtype=torch.float64, device="cuda")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            wall68[-1] = float(tt.item())
        ctx68 = sdo68.ctx
        ctx68.set_allreduce(None, 1)
        ctx68.set_solve_sharding(0, 0, None, None)
        ctx68.set_reduce_scatter(None)
        F68 = ctx68.feature_dim(0)
        T68, r68 = (F68 + 127) // 128, (((M68 + 15) // 16) * 16 + 127) // 128
        n_rows68 = int(txs68.shape[0])
        gram_exec = (T68 * (T68 + 1) // 2 + T68 * r68) * 128.0 * 128.0 * 2.0 * n_rows68      # upper 128 x 128 tiles + RHS tile columns
        gram_ms = timing68["gram"][0] / n_levels
        rcr68 = {
            "train": {
                "workload": "BASELINE config 5: RCR-68 (iBUG-68 layout, F=%d, M=%d) train, %d rows over %d GPU(s), 4 cascade levels, "
                            "MatrixNorm 1.5, synthetic 256x256 faces" % (F68, M68, n_train_global, world),
                "rows_total": int(n_train_global), "rows_per_gpu": n_rows68,
                "sec_per_cascade": wall68[-1] / n_levels, "scaling": "strong",
                "collective": ("one reduce-scatter of the owned tile columns of {A^T A, A^T b} + an all-reduce of F + 1 floats per level; " + via
                               if shard68 else "one all-reduce of {A^T A, A^T b} per level; " + via if use_dist else "none (1 GPU)"),
                "solve": ("sharded over the ranks by tile column" if shard68 else "replicated on every rank" if use_dist else "single GPU"),
                "stage_ms_per_level_rank0": {k: v[0] / n_levels for k, v in timing68.items()},
                "gram": gram_report(gram_exec, n_rows68 * float(F68) * (F68 + 1) + 2.0 * n_rows68 * F68 * M68, gram_ms),
                "solve_ms": (timing68["factor_solve"][0] + timing68["backsolve"][0]) / n_levels,
                "nlsr_per_level_rank0": list(nlsr68),
            }}
        return rcr68, ctx68, F68, L68, M68, sdo68

    # With ONE GPU the leg runs here, as in every round so far.  With several it runs BEHIND the headline measurement and under a
    # watchdog (below): its exchange -- a reduce-scatter + 213 broadcasts + 54 all-gathers per level through RCCL -- has only ever
    # run with ranks as threads on one GPU (tests/test_gpu_sharded_solve.py), and a collective that hangs on real links must not
    # take the headline of the scaling run with it.
    rcr68 = None
    if args.rcr68_shard > 0 and not defer68:
        rcr68, ctx68, F68, L68, M68, sdo68 = train_rcr68()

    # ---- workload: this rank's shard of synthetic faces, resident in HBM --------------------------------
    x_star, x0, _ = synth.make_samples(boxes[:args.batch], gt[:args.batch], ids, 0, seed=synth.SEED + 17 * rank + 1)
    d_images = torch.from_numpy(images).cuda()
    d_x0 = torch.from_numpy(x0).cuda()
    ctx.set_model_geometry(L, re, le, params)
    ctx.set_images_device(d_images.data_ptr(), args.batch, 256, 256, 256)
    ctx.set_sample_image_index(None)
    for l in range(n_levels):
        ctx.set_regressor(l, regressors[l])

    def step():
        ctx.set_x_device(d_x0.data_ptr(), args.batch)     # 720 KB device-to-device: hand the batch over
        ctx.detect_batch(fetch=False)

    # ---- algorithmic HBM bytes of the HOG launches: per level, per face,
    #      L*(2h)^2 ROI bytes + F*4 feature bytes written + 2L*4 landmark bytes read ------------------------
    ctx.enable_timing(False)
    ctx.set_x_device(d_x0.data_ptr(), args.batch)
    hog_bytes = 0          # SURVEY.md 8d per level: patch bytes + feature row written + landmarks read (the unfused launch)
    patch_bytes = 0        # ... the patch bytes alone
    for l in range(n_levels):
        ctx.hog_features(l)
        h = ctx.patch_indices()[:, 0].astype(np.int64)
        patch_bytes += int((L * (2 * h) ** 2).sum())
        hog_bytes += int((L * (2 * h) ** 2).sum()) + args.batch * (ctx.feature_dim(l) * 4 + M * 4)
        ctx.apply(l)
    x_final = ctx.get_x()
    apply_flops = sum(2.0 * args.batch * ctx.feature_dim(l) * M for l in range(n_levels))
    # sdm_detect_batch is fused (the descriptors are multiplied by the regressor on the chip): SURVEY 8d -- "if HOG is fused with the
    # apply GEMM, the feature write term drops out and is reported as such".  Algorithmic bytes of a level = patch bytes + the
    # landmark rows read and written.
    fused_bytes = patch_bytes + n_levels * args.batch * 2 * M * 4
    from superviseddescent_amd.engine import hog_plan

    def cut_fraction(cell, nl=L):      # share of the landmarks whose patch is cut by a pass boundary of the packed launch (two parts of cells)
        pl = hog_plan(5, cell, 4, nl)
        if pl is None:
            return 0.0
        def cuts(passes):
            seen = {}
            for pt in passes:
                for d in pl["lane_tab"][pt]:
                    if (int(d) >> 17) & 1:
                        seen.setdefault(int(d) & 0xff, set()).add(pt)
            return sum(1 for v in seen.values() if len(v) > 1)
        return (pl["n_main"] * cuts(range(pl["P"])) + cuts(range(pl["P"], pl["P"] + pl["Pt"]))) / float(nl)
    cut_frac = float(np.mean([cut_fraction(p.cell_size) for p in params]))

    # ---- W untimed warm-up steps, then exactly K timed steps between barrier + synchronize on both sides.  (Rounds 2-4 ran 40
    # untimed "pre-roll" steps first; round 4's own A/B showed no effect -- 3.61 M against 3.55 M faces/s without it -- so it is gone.)
    for _ in range(args.warmup):
        step()
    ctx.enable_timing(True)
    ctx.get_timing(reset=True)
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    timing = ctx.get_timing(reset=True)
    ctx.enable_timing(False)

    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # ---- the same step loop for at least one second of timed work (the 20-step headline window is 23 ms and moves with
    # the clock the chip happens to hold): steps of the headline's kind until >= 1 s has passed, the sustained rate and the clock
    # the GPU reports at its end (rocm-smi; None when the tool is not there) --------------------------------------------------------
    sustained = None
    if rank == 0:
        n_sus = max(args.steps, int(1.25 / max(dt / args.steps, 1e-6)))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n_sus):
            step()
        torch.cuda.synchronize()
        dt_sus = time.perf_counter() - t0
        sclk = None
        try:
            import re as _re
            import subprocess
            smi = subprocess.run(["rocm-smi", "-d", str(local_rank), "--showclocks"], capture_output=True, text=True, timeout=20).stdout
            m_ = _re.search(r"sclk clock level[^\n]*\((\d+)Mhz\)", smi)
            sclk = int(m_.group(1)) if m_ else None
        except Exception:      # noqa: BLE001
            pass
        sustained = {"value": args.batch * n_sus / dt_sus, "unit": "faces/s", "steps": n_sus, "seconds": dt_sus, "ms_per_step": dt_sus / n_sus * 1e3,
                     "sclk_mhz_after": sclk, "ratio_to_headline_window": (args.batch * n_sus / dt_sus) / (args.batch * args.steps / dt),
                     "note": "the headline's step loop run for >= 1 s on rank 0 (one GPU's rate): `value` at the top is the driver's "
                             "--steps window, this is the rate the chip sustains"}

    # ---- input-inclusive rate: every batch's 256 x 256 images cross PCIe (pinned host memory -> HBM) and
    # the uploads are double buffered against the cascade of the previous batch: a copy stream fills one of two device buffers while
    # the compute stream runs the cascade on the other.  The headline `value` stays the resident-input rate.
    e2e = None
    if rank == 0:
        try:
            h_images = torch.from_numpy(images[:args.batch]).pin_memory()
            h_x0 = torch.from_numpy(x0).pin_memory()
            # ONE compute stream (the cascades of consecutive batches never overlap each other: per-kernel durations stay what the
            # headline measures) + one copy stream; two device buffers, two contexts bound to them on the compute stream
            s_compute, s_copy = torch.cuda.Stream(), torch.cuda.Stream()
            bufs = [torch.empty_like(d_images[:args.batch]), torch.empty_like(d_images[:args.batch])]
            dx = [torch.empty_like(d_x0), torch.empty_like(d_x0)]
            copied = [torch.cuda.Event(), torch.cuda.Event()]
            consumed = [torch.cuda.Event(), torch.cuda.Event()]
            ctxs = []
            for k in range(2):
                c2 = Context(local_rank, stream=s_compute.cuda_stream)
                c2.set_model_geometry(L, re, le, params)
                c2.set_images_device(bufs[k].data_ptr(), args.batch, 256, 256, 256)
                c2.set_sample_image_index(None)
                for l in range(n_levels):
                    c2.set_regressor(l, regressors[l])
                ctxs.append(c2)
                consumed[k].record(s_compute)

            def e2e_step(i):
                k = i & 1
                with torch.cuda.stream(s_copy):
                    s_copy.wait_event(consumed[k])             # the cascade that last read this buffer is done
                    bufs[k].copy_(h_images, non_blocking=True)
                    dx[k].copy_(h_x0, non_blocking=True)
                    copied[k].record(s_copy)
                with torch.cuda.stream(s_compute):
                    s_compute.wait_event(copied[k])
                    ctxs[k].set_x_device(dx[k].data_ptr(), args.batch)
                    ctxs[k].detect_batch(fetch=False)
                    consumed[k].record(s_compute)
            n_e2e = max(4, min(args.steps, 24))
            for i in range(4):
                e2e_step(i)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(n_e2e):
                e2e_step(i)
            torch.cuda.synchronize()
            dt_e2e = time.perf_counter() - t0
            e2e = {"value": args.batch * n_e2e / dt_e2e, "unit": "faces/s", "ms_per_step": dt_e2e / n_e2e * 1e3, "steps": n_e2e,
                   "h2d_bytes_per_step": int(h_images.numel() + h_x0.numel() * 4),
                   "h2d_gb_per_s": (h_images.numel() + h_x0.numel() * 4) * n_e2e / dt_e2e / 1e9,
                   "note": "pinned host images -> HBM per batch on a copy stream, double buffered against the cascade of the previous "
                           "batch on the compute stream; bound by the host link (PCIe Gen5 x16, 63 GB/s spec), not by the cascade"}
            for c2 in ctxs:
                c2.close()
            del bufs, dx, h_images
        except Exception as exc:      # (reported, never fatal for the headline)
            e2e = {"error": repr(exc)}

    # ---- several GPUs: the RCR-68 legs now, under a watchdog.  If they have not finished after RCR68_DEADLINE_S seconds every rank
    # leaves (os._exit) and rank 0 prints the headline line it already has, with the legs marked as timed out.
    watchdog = None
    if defer68:
        import threading
        ms_e = dt / args.steps * 1e3
        hog_e = timing["hog"][0] / max(timing["hog"][1], 1)
        emergency = {
            "metric": "faces/sec RCR-22 detect (batch 4096)", "value": args.batch * world * args.steps / dt, "unit": "faces/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_e, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "RCR-22 detect, batch %d synthetic 256x256 u8 faces per GPU, 4 cascade levels" % args.batch,
                       "batch_per_gpu": args.batch, "levels": n_levels, "sharding": "faces sharded by rank, no collective on the detect path"},
            "roofline": {"bound": "hbm", "achieved": (fused_bytes / n_levels) / (hog_e * 1e-3) / 1e9 if hog_e > 0 else 0.0, "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": ((fused_bytes / n_levels) / (hog_e * 1e-3) / 1e9 / HBM_PEAK_GBS) if hog_e > 0 else 0.0,
                         "traffic": None, "avg_launch_ms": hog_e},
            "cpu_baseline": None,
            "train": {"error": "the legs with collectives (RCR-22 training exchange, RCR-68 training / detect) did not finish within %g s; the line was emitted by the watchdog" % RCR68_DEADLINE_S},
            "rcr68_train": {"error": "the legs with collectives did not finish within %g s; the line was emitted by the watchdog" % RCR68_DEADLINE_S},
        }

        def _expired():
            if rank == 0:
                print(json.dumps(emergency), flush=True)
            os._exit(0)
        watchdog = threading.Timer(RCR68_DEADLINE_S, _expired)
        watchdog.daemon = True
        watchdog.start()
        t0 = time.time()
        # the secondary metric "train sec/cascade" with its exchange, on a context of its own (the detect context keeps its images and
        # the model the headline ran with)
        sdo_d = SupervisedDescentOptimiser([LinearRegressor(reg()) for _ in params], device=local_rank, stream=stream)
        hog_d = HogTransform(timg, params, ids, ibug.RIGHT_EYE_IDS, ibug.LEFT_EYE_IDS, tidx, images_resident=True)
        nlsr, train_wall, train_timing = train_rcr22(True, sdo_d, hog_d)
        train_s = time.time() - t0
        sdo_d.ctx.close()
        if args.rcr68_shard > 0:
            rcr68, ctx68, F68, L68, M68, sdo68 = train_rcr68()

    # ---- BASELINE config 4: RCR-68 detect on this rank's shard (65 536 faces over 8 GPUs = 8 192 per GPU), the cascade just
    # trained, inputs resident; same timing discipline as the headline (barrier + synchronize, max over ranks) ------------------
    if rcr68 is not None:
        nb68 = args.rcr68_shard
        _, x068, _ = synth.make_samples(boxes[:nb68], gt[:nb68], ids68, 0, seed=synth.SEED + 17 * rank + 2)
        d_x068 = torch.from_numpy(x068).cuda()
        ctx68.set_images_device(d_images.data_ptr(), nb68, 256, 256, 256)
        ctx68.set_sample_image_index(None)
        ctx68.set_templates(None)
        ctx68.enable_timing(False)
        ctx68.set_x_device(d_x068.data_ptr(), nb68)
        hog_bytes68 = 0      # 2L = 136 > 64: this cascade runs through the feature matrix (csrc/sdm_capi_detect.hip fused_ok): SURVEY 8d's full byte count
        idx68_levels = []      # (half-width and cvRound'ed centres per level: the integer decisions of the parity block)
        for l in range(n_levels):
            ctx68.hog_features(l)
            idx68_levels.append(ctx68.patch_indices())
            h = idx68_levels[-1][:, 0].astype(np.int64)
            hog_bytes68 += int((L68 * (2 * h) ** 2).sum()) + nb68 * (ctx68.feature_dim(l) * 4 + M68 * 4)
            ctx68.apply(l)
        x68_stepwise = ctx68.get_x()
        steps68 = max(3, min(args.steps, 20))

        def step68():
            ctx68.set_x_device(d_x068.data_ptr(), nb68)
            ctx68.detect_batch(fetch=False)
        for _ in range(2):
            step68()
        ctx68.enable_timing(True)
        ctx68.get_timing(reset=True)
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps68):
            step68()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        dt68 = time.perf_counter() - t0
        tm68 = ctx68.get_timing(reset=True)
        ctx68.enable_timing(False)
        if use_dist:
            t = torch.tensor([dt68], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt68 = float(t.item())
        hog68_ms = tm68["hog"][0] / max(tm68["hog"][1], 1)
        app68_ms = tm68["apply"][0] / max(tm68["apply"][1], 1)
        gbs68 = hog_bytes68 / n_levels / (hog68_ms * 1e-3) / 1e9 if hog68_ms > 0 else 0.0
        tf68 = 2.0 * nb68 * F68 * M68 / (app68_ms * 1e-3) / 1e12 if app68_ms > 0 else 0.0
        rcr68["detect_shard"] = {
            "workload": "BASELINE config 4: RCR-68 detect (F=%d, M=%d), %d synthetic 256x256 faces per GPU (65 536 / 8), 4 cascade levels, "
                        "the cascade trained above" % (F68, M68, nb68),
            "value": nb68 * world * steps68 / dt68, "unit": "faces/s", "n_gpus": world, "steps": steps68,
            "ms_per_step": dt68 / steps68 * 1e3, "scaling": "weak",
            "hog": {"bound": "hbm", "achieved": gbs68, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs68 / HBM_PEAK_GBS,
                    "avg_launch_ms": hog68_ms, "algorithmic_bytes_per_launch": hog_bytes68 / n_levels},
            "apply_gemm": apply_report(2.0 * nb68 * F68 * M68, 4.0 * nb68 * F68, app68_ms),
            "path": "feature matrix + apply GEMM (the fused descriptor + apply launch serves 2L <= 64: at 2L = 136 its workgroups re-read "
                    "230 KB of regressor per 32 faces from L2 -- measured 0.53 against 0.25 ms per level)",
        }
        x68_fused = ctx68.get_x()      # (the last timed step's landmarks)
    if watchdog is not None:
        watchdog.cancel()

    if rank != 0:
        if lazy_rccl is not None:
            lazy_rccl.destroy()
        if use_dist:
            dist.destroy_process_group()
        return

    ms_per_step = dt / args.steps * 1e3
    faces_per_s = args.batch * world * args.steps / dt
    hog_ms, hog_n = timing["hog"]
    app_ms, app_n = timing["apply"]
    hog_avg_ms = hog_ms / max(hog_n, 1)
    bytes_per_launch = fused_bytes / n_levels          # patch bytes + landmark rows: the feature write dropped out (fused)
    achieved_gbs = bytes_per_launch / (hog_avg_ms * 1e-3) / 1e9 if hog_avg_ms > 0 else 0.0

    # HBM bytes and instruction counts per launch from the PMC passes (rocprofv3 --pmc cannot run inside this process): the
    # committed measurement of this same command (scripts/profile_bench.sh), valid for the batch it was taken at
    traffic, traffic_src, valu_issue, traffic_by_kernel = None, None, None, None
    hog_kernel = "hog_packed_kernel"
    try:
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "hbm_traffic.json")) as fh:
            tj = json.load(fh)
        ent = tj["kernels"][hog_kernel]
        if int(tj.get("batch", 0)) == args.batch:
            # the PAIR that replaced round 3's single launch: the pixel kernel writes raw cell histograms, the
            # descriptor kernel reads them back -- that round trip is part of what a level moves through the memory system
            traffic_by_kernel = {k: float(tj["kernels"][k]["bytes_per_launch"]) for k in (hog_kernel, "desc_kernel") if k in tj["kernels"]}
            traffic, traffic_src = sum(traffic_by_kernel.values()), "committed-profile: profiles/hbm_traffic.json (" + tj["source"] + ")"
            # vector-unit occupancy of the launch from the same PMC passes: SQ_ACTIVE_INST_VALU counts quad-cycles summed over the
            # waves; x 4 / (1024 SIMDs x launch cycles) = the fraction of SIMD cycles with a vector instruction executing
            simd_cycles = 256 * 4 * 2.4e9 * hog_avg_ms * 1e-3
            valu_issue = {"valu_insts_per_launch": float(ent["SQ_INSTS_VALU"]), "salu_insts_per_launch": float(ent.get("SQ_INSTS_SALU", 0.0)),
                          "mfma_f32_busy_frac": (float(ent["SQ_VALU_MFMA_BUSY_CYCLES"]) / simd_cycles) if "SQ_VALU_MFMA_BUSY_CYCLES" in ent else None,
                          "lds_busy_frac": (float(ent["SQ_LDS_IDX_ACTIVE"]) / (256 * 2.4e9 * hog_avg_ms * 1e-3)) if "SQ_LDS_IDX_ACTIVE" in ent else None,
                          "simd_cycles_per_valu_inst": simd_cycles / float(ent["SQ_INSTS_VALU"]),
                          "valu_insts_per_patch": float(ent["SQ_INSTS_VALU"]) / (args.batch * L),
                          "frac": (4.0 * float(ent["SQ_ACTIVE_INST_VALU"]) / simd_cycles) if "SQ_ACTIVE_INST_VALU" in ent else None,
                          "lds_bank_conflict_ratio": (float(ent["SQ_LDS_BANK_CONFLICT"]) / float(ent["SQ_LDS_IDX_ACTIVE"]))
                          if "SQ_LDS_IDX_ACTIVE" in ent and ent["SQ_LDS_IDX_ACTIVE"] else None,
                          "note": "counter view of the same launch, from the committed PMC passes of this command: frac = 4 x "
                                  "SQ_ACTIVE_INST_VALU / (1024 SIMDs x 2.4 GHz x this run's average launch time) = share of SIMD cycles with a "
                                  "vector instruction executing (the vector pipe is the busy unit); mfma_f32_busy_frac = the band folds' matrix "
                                  "instructions; lds_busy_frac = LDS array cycles.  Consistent with `compute`: the launch is bound by "
                                  "vector-instruction issue; the ablations of profiles/r04_hog_ablations.txt price WHICH instructions "
                                  "(image-load path 24 %, band folds 20 %, per-row LDS table reads 11 %, binning 8 %, resize arithmetic 4 %)"}
    except (OSError, KeyError, ValueError, ZeroDivisionError):
        pass

    # compute-side roofline of the same launch: what the vector-instruction issue alone would take (profiles/r04_issue_model.json,
    # scripts/isa_issue_model.py: per-class instruction counts of the unrolled pass from the ISA x the issue clocks measured on
    # this chip, + the vector clocks lost while the band folds' matrix instructions run on the same SIMD)
    compute_roof = None
    try:
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r04_issue_model.json")) as fh:
            im = json.load(fh)
        cells = [p[2] for p in ibug.SHIPPED_HOG_PARAMS]
        if all(str(c) in im["levels"] for c in cells) and L == 22:
            pred = []
            for c in cells:
                e = im["levels"][str(c)]
                clocks = e["valu_clocks_row_loop_and_folds"] + e["valu_clocks_lost_beside_mfma_per_pass"]      # (set-up not counted: a lower bound)
                pred.append(args.batch * e["passes_per_face"] * clocks / 1024.0 / 2.4e9 * 1e3)
            pred_ms = sum(pred) / len(pred)
            compute_roof = {"bound": "valu_issue", "predicted_avg_launch_ms": pred_ms, "measured_avg_launch_ms": hog_avg_ms,
                            "frac": pred_ms / hog_avg_ms if hog_avg_ms > 0 else None,
                            "valu_clocks_per_pixel_row": [im["levels"][str(c)]["valu_clocks_per_row"] for c in cells],
                            "note": "frac = (time the pass's vector instructions need at their measured issue rates, 1 024 SIMDs at 2.4 GHz) / "
                                    "(measured launch time): the launch runs at this fraction of its instruction-issue bound; the rest is per-wave "
                                    "set-up, the tail of the launch (6-12 % at this batch, scripts/r4_scaling.py) and stalls.  source: "
                                    "profiles/r04_issue_model.json"}
    except (OSError, KeyError, ValueError, ZeroDivisionError):
        pass

    F22 = ctx.feature_dim(0)
    T22 = (F22 + 127) // 128
    pair_ms = hog_avg_ms + app_ms / max(app_n, 1)
    out = {
        "metric": "faces/sec RCR-22 detect (batch 4096)",
        "value": faces_per_s,
        "unit": "faces/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": "RCR-22 detect, batch %d synthetic 256x256 u8 faces per GPU, 4 cascade levels "
                        "(UoCTTI HOG 5x5 cells, cell 11/10/8/6, 4 orientations, F=8801, M=44), model trained "
                        "on-GPU on %d synthetic faces (shipped .bin absent from the reference checkout)"
                        % (args.batch, n_train_global),
            "batch_per_gpu": args.batch,
            "levels": n_levels,
            "sharding": "faces sharded by rank, no collective on the detect path",
        },
        "roofline": {
            "kernel": hog_kernel,
            "bound": "hbm",
            "achieved": achieved_gbs,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": achieved_gbs / HBM_PEAK_GBS,
            "traffic": traffic,
            "traffic_by_kernel": traffic_by_kernel,
            "traffic_source": traffic_src,
            "pair": {"kernels": "hog_packed_kernel + desc_kernel<FUSED> + apply_reduce_kernel (one cascade level of sdm_detect_batch)",
                     "avg_level_ms": pair_ms,
                     "achieved": bytes_per_launch / (pair_ms * 1e-3) / 1e9 if pair_ms > 0 else 0.0,
                     "frac": bytes_per_launch / (pair_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if pair_ms > 0 else 0.0,
                     "traffic_over_algorithmic": (traffic / bytes_per_launch) if (traffic and bytes_per_launch) else None,
                     "note": "the same algorithmic bytes over the whole level (pixel kernel + descriptor / product / update launches): "
                             "what the round-3 single launch + GEMM was replaced by"},
            "valu_issue": valu_issue,
            "compute": compute_roof,
            "algorithmic_bytes_per_launch": bytes_per_launch,
            "feature_write": "dropped: sdm_detect_batch multiplies the descriptors by the regressor on the chip (csrc/sdm_desc.hip); the "
                             "launch writes raw cell histograms (800 B per patch, %.1f MB per launch) for that kernel instead of the "
                             "%.1f MB feature matrix -- SURVEY.md 8d: 'if fused the feature write term drops out'"
                             % (args.batch * L * 800.0 * (1.0 + cut_frac) / 1e6, sum(4.0 * args.batch * ctx.feature_dim(l) for l in range(n_levels)) / n_levels / 1e6),
            "achieved_with_round3_byte_count": (hog_bytes / n_levels) / (hog_avg_ms * 1e-3) / 1e9 if hog_avg_ms > 0 else 0.0,
            "note": "pixel kernel of the split launch (crop + cv::resize + gradient + orientation binning + column sums + band folds -> raw "
                    "cells).  Not HBM bound: `traffic` (pixel + descriptor kernel, counters) is ~1.9 x the algorithmic bytes because the raw "
                    "cells make a round trip through memory, and even that is ~15 % of what HBM delivers in the level's time; the launch is "
                    "bound by the vector instructions a pixel row must issue -- `compute` is its issue-bound roofline, valu_issue the counters",
            "avg_launch_ms": hog_avg_ms,
            "launches": hog_n,
        },
        "fused_apply": fused_apply_report(args.batch, L, 400, M, cut_frac, app_ms / max(app_n, 1)),
        "e2e_with_h2d": e2e,
        "train": {
            "metric": "train sec/cascade (RCR-22, MatrixNorm 1.5, bias unregularised)",
            "rows_total": int(n_train_global),
            "rows_per_gpu": int(txs.shape[0]),
            "sec_per_cascade": train_wall[-1] / n_levels,
            "scaling": "strong",
            "collective": ("one reduce-scatter of the owned tile columns of {A^T A, A^T b} + an all-reduce of F + 1 floats per level; " + via
                           if shard_solve else "one all-reduce of {A^T A, A^T b} per level; " + via if use_dist else "none (1 GPU)"),
            "solve": ("sharded over the ranks by tile column: one <= 4-tile broadcast per 128-column step, one all-gather per 4 steps"
                      if shard_solve else "replicated on every rank" if use_dist else "single GPU"),
            "stage_ms_per_level_rank0": {k: v[0] / n_levels for k, v in train_timing.items()},
            "gram": gram_report((T22 * (T22 + 1) // 2 + T22) * 128.0 * 128.0 * 2.0 * int(txs.shape[0]),
                                int(txs.shape[0]) * float(F22) * (F22 + 1) + 2.0 * int(txs.shape[0]) * F22 * M,
                                train_timing["gram"][0] / n_levels),
            "nlsr_per_level_rank0": nlsr,
            "seconds_total_two_passes": train_s,
            "seconds_data_generation": datagen_s,
        },
    }

    if rcr68 is not None:
        out["rcr68_train"] = rcr68["train"]
        out["rcr68_detect_shard"] = rcr68["detect_shard"]
    out["sustained"] = sustained

    # ---- BASELINE config 3 at its stated shape (VERDICT r05 row g): RCR-22, 5 cascade levels, 31-dimensional VlHog (9 orientations:
    # F = 22 x 25 x 31 + 1 = 17 051), ridge lambda = 1.0 (Manual), 10 000 rows -- the first 10 000 training rows of this rank, one GPU,
    # no collective; the second of two passes is timed ------------------------------------------------------------------------------
    try:
        p3 = [HoGParam(*p) for p in [(1, 5, 11, 9, 1.0), (1, 5, 10, 9, 0.7), (1, 5, 8, 9, 0.4), (1, 5, 6, 9, 0.25), (1, 5, 6, 9, 0.25)]]
        n3 = min(10000, int(txs.shape[0]))
        n3_img = int(tidx[:n3].max()) + 1
        sdo3 = SupervisedDescentOptimiser([LinearRegressor(Regulariser(Regulariser.RegularisationType.Manual, 1.0, True)) for _ in p3],
                                          device=local_rank, stream=stream)
        hog3 = HogTransform(timg[:n3_img], p3, ids, ibug.RIGHT_EYE_IDS, ibug.LEFT_EYE_IDS, tidx[:n3], images_resident=True)
        nlsr3 = []
        for rep in range(2):
            sdo3.ctx.enable_timing(True)
            sdo3.ctx.get_timing(reset=True)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            sdo3.train(txs[:n3], tx0[:n3], None, hog3,
                       on_training_epoch_callback=(lambda cur: nlsr3.append(float(np.linalg.norm(cur - txs[:n3]) / np.linalg.norm(txs[:n3])))) if rep == 0 else None)
            torch.cuda.synchronize()
            wall3 = time.perf_counter() - t1
            tm3 = sdo3.ctx.get_timing(reset=True)
        F3 = sdo3.ctx.feature_dim(0)
        T3 = (F3 + 127) // 128
        out["config3_train"] = {
            "workload": "BASELINE config 3: RCR-22 train, 5 cascade levels, 31-dimensional VlHog (9 orientations, 5 x 5 cells of 11/10/8/6/6, "
                        "F = %d, M = %d), %d synthetic faces (rows), ridge lambda = 1.0 (Manual, every row regularised), 1 GPU" % (F3, M, n3),
            "rows": n3, "levels": len(p3), "sec_per_cascade": wall3 / len(p3), "seconds_total": wall3,
            "stage_ms_per_level": {k: v[0] / len(p3) for k, v in tm3.items() if v[1] > 0},
            "gram": gram_report((T3 * (T3 + 1) // 2 + T3) * 128.0 * 128.0 * 2.0 * n3, n3 * float(F3) * (F3 + 1) + 2.0 * n3 * F3 * M,
                                tm3["gram"][0] / len(p3)),
            "solve_ms": (tm3["factor_solve"][0] + tm3["backsolve"][0]) / len(p3),
            "nlsr_per_level": nlsr3,
            "note": "parity of this configuration: tests/test_gpu_configs.py (teacher-forced against the oracle at every level)"}
        sdo3.ctx.close()
    except Exception as exc:      # (reported, never fatal for the headline)
        out["config3_train"] = {"error": repr(exc)}

    # ---- CPU baseline: the oracle on this box's host cores, bounded sample of the same batch -----------
    if not args.no_cpu:
        from oracle import sdm_oracle as orc
        from superviseddescent_amd import _lib
        cores = os.cpu_count() or 1
        ref_hog = orc.use_reference_hog(True)      # the reference's own include/rcr/hog.c (oracle/_ref), when present
        # bounded sample: ~5 ms of CPU work per face and level-set => about 20 core-seconds in total
        ns = min(args.cpu_sample or max(256, min(4096, 16 * cores)), args.batch)
        oparams = [orc.HoGParam(*p) for p in ibug.SHIPPED_HOG_PARAMS]
        oregs = []
        for l in range(n_levels):
            r = orc.LinearRegressor()
            r.x = regressors[l]
            oregs.append(r)
        osdo = orc.SupervisedDescentOptimiser(oregs, orc.InterEyeDistanceNormalisation(re, le))
        ohog = orc.HogTransform(images[:ns], oparams, re, le, None, n_threads=cores)
        ohog.keep_idx = True
        t0 = time.perf_counter()
        ox = osdo.test(x0[:ns], None, ohog)
        cpu_dt = time.perf_counter() - t0
        n1 = min(128, ns)
        ohog1 = orc.HogTransform(images[:n1], oparams, re, le, None, n_threads=1)
        t0 = time.perf_counter()
        osdo.test(x0[:n1], None, ohog1)
        cpu1_dt = time.perf_counter() - t0
        out["cpu_baseline"] = {
            "value": ns / cpu_dt,
            "unit": "faces/s",
            "cores": cores,
            "kind": "reference-hog" if ref_hog else "port",
            "sample": "first %d faces of the batch, full 4-level cascade: the reference's HogTransform restated (crop, cv::resize 8U, "
                      "reorder) around %s, task-per-sample on %d threads with a pool per level as in superviseddescent.hpp:173-177; "
                      "single thread: %.1f faces/s on %d faces"
                      % (ns, "the reference's own hog.c (oracle/_ref/libref_hog.so)" if ref_hog else "the restated hog.c", cores,
                         n1 / cpu1_dt, n1),
            "single_thread_value": n1 / cpu1_dt,
        }

        # parity on the sample, free-running: landmarks, integer decisions per level, per-face errors -- in the default
        # accumulation mode and in SDM_HOG_EXACT_ORDER (features bit-identical to the reference's raster-order sums)
        def gpu_cascade(mode):
            ctx.set_hog_mode(mode)
            ctx.set_images_device(d_images.data_ptr(), ns, 256, 256, 256)
            ctx.set_x(x0[:ns])
            idxs = []
            for l in range(n_levels):
                ctx.hog_features(l)
                idxs.append(ctx.patch_indices())
                ctx.apply(l)
            return ctx.get_x(), idxs

        def compare(xg, idxs):
            diverged = np.zeros(ns, bool)
            for l in range(n_levels):
                diverged |= (idxs[l] != ohog.idx_per_level[l]).any(axis=1)
            d = (xg - ox).astype(np.float64)
            per_face = np.linalg.norm(d, axis=1) / np.linalg.norm(ox.astype(np.float64), axis=1)
            same = ~diverged
            return {"rel_l2_landmarks_vs_oracle": float(np.linalg.norm(d) / np.linalg.norm(ox.astype(np.float64))),
                    "faces_with_different_integer_decisions": int(diverged.sum()),
                    "max_per_face_rel_error": float(per_face.max()),
                    "max_per_face_rel_error_same_decisions": float(per_face[same].max()) if same.any() else None,
                    "rel_l2_same_decisions": float(np.linalg.norm(d[same]) / np.linalg.norm(ox[same].astype(np.float64))) if same.any() else None}

        par = compare(*gpu_cascade(_lib.SDM_HOG_COLUMNS))
        par_exact = compare(*gpu_cascade(_lib.SDM_HOG_EXACT_ORDER))
        ctx.set_hog_mode(_lib.SDM_HOG_COLUMNS)
        # the MEASURED path: sdm_detect_batch (fused descriptor + apply launches, no feature matrix) on the same sample
        ctx.set_images_device(d_images.data_ptr(), ns, 256, 256, 256)
        ctx.set_x(x0[:ns])
        x_fused = ctx.detect_batch(fetch=True)

        def compare_x(xg, xo):
            d = (xg - xo).astype(np.float64)
            per_face = np.linalg.norm(d, axis=1) / np.linalg.norm(xo.astype(np.float64), axis=1)
            return {"rel_l2_landmarks_vs_oracle": float(np.linalg.norm(d) / np.linalg.norm(xo.astype(np.float64))),
                    "max_per_face_rel_error": float(per_face.max()), "faces_above_1e-4": int((per_face > 1e-4).sum())}
        # The oracle's predict restates cv::gemm's DOUBLE accumulation (SURVEY a-6; the default of oracle/sdm_oracle.py since round 5).
        # The labelled alternative -- a float32-accumulating BLAS sgemm, what an OpenCV built on a BLAS back-end would run -- beside it:
        oregs32 = []
        for l in range(n_levels):
            r = orc.LinearRegressor(accumulate_double=False)
            r.x = regressors[l]
            oregs32.append(r)
        ohog32 = orc.HogTransform(images[:ns], oparams, re, le, None, n_threads=cores)
        ohog32.keep_idx = True
        ox32 = orc.SupervisedDescentOptimiser(oregs32, orc.InterEyeDistanceNormalisation(re, le)).test(x0[:ns], None, ohog32)
        idx_keep = ohog.idx_per_level
        ohog.idx_per_level = ohog32.idx_per_level
        ox_keep, ox = ox, ox32
        par32 = compare(*gpu_cascade(_lib.SDM_HOG_COLUMNS))
        ox, ohog.idx_per_level = ox_keep, idx_keep
        out["parity"] = dict(par, faces_checked=ns, tolerance=1e-4, levels=n_levels,
                             oracle_predict="double-accumulating (cv::gemm, SURVEY a-6)",
                             exact_order_mode=par_exact,
                             detect_batch_fused=dict(compare_x(x_fused, ox), vs_sgemm_accumulating_oracle=compare_x(x_fused, ox32),
                                                     note="sdm_detect_batch as timed (descriptors x regressor slices on the chip)"),
                             vs_sgemm_accumulating_oracle=par32,
                             note="free-running 4-level cascade; a face 'differs in integer decisions' when a cvRound'ed patch centre or the "
                                  "patch half-width at any level differs from the oracle's (a landmark within float rounding of x.5): from "
                                  "there on it is a different, equally valid trajectory.  The oracle's predict accumulates in double and rounds "
                                  "to float, as cv::gemm does; vs_sgemm_accumulating_oracle = the same comparison against a float32-accumulating "
                                  "sgemm (the oracle's default until round 4; its own rounding moves a few knife-edge faces)")

        # ---- RCR-68 (BASELINE configs 4 / 5): the detect shard's first faces against the oracle, regressors as trained on the GPU ----
        if rcr68 is not None:
            # the WHOLE shard (2 knife-edge faces among the first 512 say little about 8 192): ~3 x the RCR-22 oracle's work
            # per face; a box with few cores checks a prefix sized for ~40 core-minutes and says so
            n68 = nb68 if cores >= 32 else min(nb68, max(512, 40 * cores))
            re68, le68 = ibug.eye_indices(ids68)
            oregs68 = []
            for r68_ in sdo68.regressors:
                r = orc.LinearRegressor()
                r.x = r68_.x
                oregs68.append(r)
            ohog68 = orc.HogTransform(images[:n68], oparams, re68, le68, None, n_threads=cores)
            ohog68.keep_idx = True
            t0 = time.perf_counter()
            ox68 = orc.SupervisedDescentOptimiser(oregs68, orc.InterEyeDistanceNormalisation(re68, le68)).test(x068[:n68], None, ohog68)
            cpu68_dt = time.perf_counter() - t0
            div68 = np.zeros(n68, bool)
            for l in range(n_levels):
                div68 |= (idx68_levels[l][:n68] != ohog68.idx_per_level[l]).any(axis=1)
            d68 = (x68_fused[:n68] - ox68).astype(np.float64)
            pf68 = np.linalg.norm(d68, axis=1) / np.linalg.norm(ox68.astype(np.float64), axis=1)
            out["rcr68_detect_shard"]["parity"] = dict(compare_x(x68_fused[:n68], ox68), faces_checked=n68, tolerance=1e-4,
                                                       faces_with_different_integer_decisions=int(div68.sum()),
                                                       faces_above_1e_4_with_same_decisions=int(((pf68 > 1e-4) & ~div68).sum()),
                                                       max_per_face_rel_error_same_decisions=float(pf68[~div68].max()) if (~div68).any() else None,
                                                       stepwise_unfused=compare_x(x68_stepwise[:n68], ox68),
                                                       cpu_oracle_faces_per_s=n68 / cpu68_dt, cores=cores,
                                                       whole_shard=bool(n68 == nb68),
                                                       note="free-running 4-level RCR-68 cascade (F = 27 201, M = 136), sdm_detect_batch of the "
                                                            "timed shard against the CPU oracle" + (" on every face of the shard" if n68 == nb68 else
                                                            " on its first %d faces (%d host cores: the whole shard would take ~%d core-minutes)" % (n68, cores, int(nb68 * 0.3))))

        # ---- CPU training baseline (sub-sampled, per stage as verbose_solver.hpp:66-97 prints): level 0 of the RCR-22
        # cascade on the first 2 000 training rows, the oracle's stages timed on this box's cores, the GPU on the same rows ----------
        nt = min(2000, int(txs.shape[0]))
        n_timg = int(tidx[:nt].max()) + 1
        t_hog = orc.HogTransform(timg[:n_timg], oparams[:1], re, le, tidx[:nt], n_threads=cores)
        t0 = time.perf_counter(); A = np.asarray(t_hog(tx0[:nt], 0), np.float32); cpu_hog = time.perf_counter() - t0
        bvec = ((tx0[:nt] - txs[:nt]) * orc.InterEyeDistanceNormalisation(re, le)(tx0[:nt])).astype(np.float32)
        t0 = time.perf_counter(); AtA = (A.T @ A).astype(np.float32); Atb = (A.T @ bvec).astype(np.float32); cpu_gram = time.perf_counter() - t0
        oreg = orc.Regulariser(orc.Regulariser.MATRIX_NORM, 1.5, False)
        t0 = time.perf_counter()
        lam = oreg.get_lambda(AtA, nt); dg = np.full(AtA.shape[0], lam, np.float32); dg[-1] = 0.0; AtA[np.diag_indices_from(AtA)] += dg
        cpu_reg = time.perf_counter() - t0
        from scipy.linalg import lu_factor, lu_solve
        t0 = time.perf_counter(); lu = lu_factor(AtA, check_finite=False); cpu_lu = time.perf_counter() - t0
        t0 = time.perf_counter(); Rcpu = lu_solve(lu, Atb, check_finite=False).astype(np.float32); cpu_sub = time.perf_counter() - t0
        t0 = time.perf_counter(); _ = A @ Rcpu; cpu_apply = time.perf_counter() - t0
        sdo_s = SupervisedDescentOptimiser([LinearRegressor(reg())], device=local_rank, stream=stream)
        hog_s = HogTransform(timg[:n_timg], params[:1], ids, ibug.RIGHT_EYE_IDS, ibug.LEFT_EYE_IDS, tidx[:nt])
        for rep in range(2):
            sdo_s.ctx.enable_timing(True); sdo_s.ctx.get_timing(reset=True)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            sdo_s.train(txs[:nt], tx0[:nt], None, hog_s)
            torch.cuda.synchronize(); gpu_level = time.perf_counter() - t0
            tg = sdo_s.ctx.get_timing(reset=True)
        Rg = sdo_s.regressors[0].x
        # what the regressor difference means in landmarks: both regressors applied to the SAME (CPU) feature rows, one update step
        ied = 1.0 / orc.InterEyeDistanceNormalisation(re, le)(tx0[:nt]).astype(np.float64)
        x1_cpu = tx0[:nt].astype(np.float64) - (A.astype(np.float64) @ Rcpu.astype(np.float64)) * ied
        x1_gpu = tx0[:nt].astype(np.float64) - (A.astype(np.float64) @ Rg.astype(np.float64)) * ied
        rows_heldout = slice(nt, min(2 * nt, int(txs.shape[0])))
        lm_heldout = None
        if rows_heldout.stop - rows_heldout.start >= 100:
            n_h = int(tidx[rows_heldout].max()) + 1
            h_hog = orc.HogTransform(timg[:n_h], oparams[:1], re, le, tidx[rows_heldout], n_threads=cores)
            Ah = np.asarray(h_hog(tx0[rows_heldout], 0), np.float64)
            iedh = 1.0 / orc.InterEyeDistanceNormalisation(re, le)(tx0[rows_heldout]).astype(np.float64)
            xh_cpu = tx0[rows_heldout].astype(np.float64) - (Ah @ Rcpu.astype(np.float64)) * iedh
            xh_gpu = tx0[rows_heldout].astype(np.float64) - (Ah @ Rg.astype(np.float64)) * iedh
            lm_heldout = float(np.linalg.norm(xh_gpu - xh_cpu) / np.linalg.norm(xh_cpu))
        out["cpu_baseline_train"] = {
            "workload": "level 0 of the RCR-22 training cascade on the first %d training rows (sub-sampled: the full 100 000-row A^T A is "
                        "1.5 x 10^13 flop per level), MatrixNorm 1.5, bias unregularised" % nt,
            "cores": cores, "kind": "reference-hog" if ref_hog else "port", "unit": "ms",
            "cpu_stage_ms": {"hog_features": cpu_hog * 1e3, "AtA_Atb (numpy sgemm)": cpu_gram * 1e3, "regulariser": cpu_reg * 1e3,
                             "partial-pivot LU (scipy sgetrf)": cpu_lu * 1e3, "substitution": cpu_sub * 1e3, "apply": cpu_apply * 1e3},
            "cpu_total_ms": (cpu_hog + cpu_gram + cpu_reg + cpu_lu + cpu_sub + cpu_apply) * 1e3,
            "gpu_same_rows_stage_ms": {k: v[0] for k, v in tg.items() if v[1] > 0},
            "gpu_same_rows_wall_ms": gpu_level * 1e3,
            "regressor_rel_l2_gpu_vs_cpu": float(np.linalg.norm(Rg - Rcpu) / np.linalg.norm(Rcpu)),
            "landmarks_rel_l2_gpu_vs_cpu_regressor": {"training_rows": float(np.linalg.norm(x1_gpu - x1_cpu) / np.linalg.norm(x1_cpu)),
                                                      "held_out_rows": lm_heldout,
                                                      "note": "one update step with either regressor on the same feature rows: with "
                                                              "2 000 rows for 8 801 unknowns the system is under-determined up to the "
                                                              "regulariser, so two float32 solvers (LU on the CPU, Cholesky on the GPU) "
                                                              "differ in directions the data does not see -- the landmarks do not move"},
            "note": "CPU stages as the reference's VerbosePartialPivLUSolver prints them (verbose_solver.hpp:66-97): Eigen's f32 GEMM / "
                    "PartialPivLU restated with numpy sgemm / LAPACK sgetrf on all cores; never extrapolated to the full row count"}
    print(json.dumps(out))
    if lazy_rccl is not None:
        lazy_rccl.destroy()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
