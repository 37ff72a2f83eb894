"""The N-GPU training path on the real backend: torch.distributed "nccl" (= RCCL) summing the engine's Gram/RHS buffer
in place through the sdm_set_allreduce callback (zero-copy view of engine HBM on torch's stream).  One rank per
visible GPU (the test box has one), launched exactly as the bench driver launches bench.py."""
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _torchrun(nproc, script, *args, timeout=600, **extra_env):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **extra_env)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), script, *args]
    return subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)


@pytest.mark.gpu
def test_rccl_allreduce_training(built):
    import torch
    n = torch.cuda.device_count()
    assert n >= 1
    r = _torchrun(min(n, 2), os.path.join(ROOT, "tests", "_rccl_worker.py"))
    assert r.returncode == 0 and "RCCL_WORKER_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


@pytest.mark.gpu
def test_bench_under_torchrun(built):
    """bench.py launched the way the driver launches it for N > 1 (here N = visible GPUs, at most 2), reduced sizes."""
    import json
    import torch
    n = min(torch.cuda.device_count(), 2)
    r = _torchrun(n, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "2", "--warmup", "1", "--batch", "256",
                  "--train-rows", "800", "--no-cpu", SDM_BENCH_SHARD_SOLVE="1")
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == n and out["value"] > 0 and out["scaling"] == "weak"
    assert "RCCL" in out["train"]["collective"] and out["train"]["solve"].startswith("sharded")


@pytest.mark.gpu
def test_bench_several_gpu_order_and_watchdog(built):
    """With more than one GPU bench.py runs its RCR-68 legs BEHIND the headline measurement and under a watchdog that emits the
    headline line if their collectives hang.  The order is exercised here on the visible GPUs (SDM_BENCH_DEFER_RCR68=1), and the
    watchdog by a deadline of zero seconds: the line must still be one valid JSON object with the headline and the legs marked."""
    import json
    import torch
    n = min(torch.cuda.device_count(), 2)
    common = ("--gpus", str(n), "--steps", "2", "--warmup", "1", "--batch", "256", "--train-rows", "800", "--rcr68-shard", "256", "--no-cpu")
    r = _torchrun(n, os.path.join(ROOT, "bench.py"), *common, SDM_BENCH_DEFER_RCR68="1")
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["value"] > 0 and out["rcr68_train"]["sec_per_cascade"] > 0 and out["rcr68_detect_shard"]["value"] > 0
    r = _torchrun(n, os.path.join(ROOT, "bench.py"), *common, SDM_BENCH_DEFER_RCR68="1", SDM_BENCH_RCR68_DEADLINE_S="0")
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["value"] > 0 and out["n_gpus"] == n and "error" in out["rcr68_train"] and out["roofline"]["frac"] > 0


@pytest.mark.gpu
@pytest.mark.parametrize("world,xblocks", [(2, None), (3, None), (4, None), (2, 3), (3, 4)])
def test_training_collectives_with_real_processes(built, world, xblocks):
    """World sizes > 1 through the product's exchange callbacks with REAL processes: torch.distributed's gloo backend moves device
    tensors through the host, so the ranks may share the one GPU of the test box (RCCL refuses two ranks per device).  The worker
    trains with the all-reduce, with the sharded factorisation (broadcast + all-gather per step / group) and with the
    reduce-scatter exchange, and checks: identical regressors on every rank in all three forms, sharded == replicated bit for bit,
    the expected number of collectives, and the regressors (5e-5; measured 2e-5) and landmarks (1e-6; measured 2e-8) of single-process training on all rows."""
    # xblocks: the reduce-scatter exchange block-wise behind the Gram kernel (SDM_GRAM_XBLOCKS; automatic only from 128 tile columns on)
    extra = {"SDM_GRAM_XBLOCKS": str(xblocks)} if xblocks else {}
    r = _torchrun(world, os.path.join(ROOT, "tests", "_rccl_worker.py"), SDM_TEST_BACKEND="gloo", **extra)
    assert r.returncode == 0 and "RCCL_WORKER_OK world=%d" % world in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]


@pytest.mark.gpu
def test_bench_with_two_real_processes(built):
    """bench.py --gpus 2 as two real processes (gloo, sharing the GPU): the several-GPU order -- headline first, then the legs with
    collectives under the watchdog -- the RCR-22 all-reduce training, and the RCR-68 leg with the reduce-scatter exchange and the
    factorisation sharded over the two ranks."""
    import json
    r = _torchrun(2, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "512", "--train-rows", "4000",
                  "--rcr68-shard", "256", "--no-cpu", SDM_BENCH_BACKEND="gloo", timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["value"] > 0
    assert "all-reduce" in out["train"]["collective"] and out["train"]["sec_per_cascade"] > 0
    assert out["rcr68_train"]["solve"].startswith("sharded") and "reduce-scatter" in out["rcr68_train"]["collective"]
    nl = out["rcr68_train"]["nlsr_per_level_rank0"]
    assert len(nl) == 4 and all(b < a for a, b in zip(nl, nl[1:]))              # the cascade trained by two ranks converges
    assert out["rcr68_detect_shard"]["value"] > 0


@pytest.mark.gpu
def test_bench_relaunches_itself_for_several_gpus(built):
    """`python bench.py --gpus 2` with NO launcher (no RANK / WORLD_SIZE in the environment): the script replaces itself by
    `python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2 ...` and the line says n_gpus = 2 -- it used to run one
    rank silently and print n_gpus = 1.  gloo, the two ranks sharing the GPU of the test box."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0", SDM_BENCH_BACKEND="gloo")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "256",
                        "--train-rows", "800", "--rcr68-shard", "0", "--no-cpu"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["value"] > 0 and out["train"]["sec_per_cascade"] > 0


@pytest.mark.gpu
def test_bench_collectives_are_issued_by_the_library(built):
    """Under torch.distributed.run on the nccl backend bench.py's collectives go through the library's own RCCL calls
    (parallel.RcclCommunicator -> sdm_set_allreduce_rccl / sdm_set_reduce_scatter_rccl / sdm_set_solve_sharding_rccl), not through
    Python callbacks; SDM_BENCH_COLLECTIVES=torch keeps the callback path for an A/B.  One rank per visible GPU (the test box has
    one: every collective is issued, the sums are identities) -- both paths must train the same cascade."""
    import json
    import torch
    n = min(torch.cuda.device_count(), 2)
    common = ("--gpus", str(n), "--steps", "2", "--warmup", "1", "--batch", "256", "--train-rows", "800", "--rcr68-shard", "256", "--no-cpu")
    outs = {}
    for how in ("rccl", "torch"):
        r = _torchrun(n, os.path.join(ROOT, "bench.py"), *common, SDM_BENCH_COLLECTIVES=how, SDM_BENCH_SHARD_SOLVE="1")
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
        outs[how] = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert "called by the library" in outs["rccl"]["train"]["collective"] and "callbacks" in outs["torch"]["train"]["collective"]
    assert outs["rccl"]["train"]["solve"].startswith("sharded")
    for leg in ("train", "rcr68_train"):
        a, b = outs["rccl"][leg]["nlsr_per_level_rank0"], outs["torch"][leg]["nlsr_per_level_rank0"]
        assert len(a) == 4 and a == pytest.approx(b, rel=1e-5)
