"""Does torch.distributed's gloo backend move CUDA (HIP) tensors on this build?  Two ranks on ONE GPU: all_reduce, broadcast,
all_gather_into_tensor, reduce_scatter_tensor.  (If it does, bench.py's several-GPU path can be run by two real processes on the
one-GPU test box: SDM_BENCH_BACKEND=gloo.)  usage: python -m torch.distributed.run --nproc-per-node 2 scripts/gloo_gpu_probe.py"""
import os, torch, torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
dist.init_process_group("gloo")
r, w = dist.get_rank(), dist.get_world_size()
dev = torch.device("cuda", 0)
res = {}
def attempt(name, fn):
    try:
        fn(); torch.cuda.synchronize(); res[name] = "ok"
    except Exception as e:
        res[name] = "FAILED: " + repr(e)[:120]
t = torch.full((1000,), float(r + 1), device=dev)
attempt("all_reduce", lambda: dist.all_reduce(t))
b = torch.full((10,), float(r), device=dev)
attempt("broadcast", lambda: dist.broadcast(b, src=1))
g = torch.empty(20, device=dev)
attempt("all_gather_into_tensor", lambda: dist.all_gather_into_tensor(g, torch.full((10,), float(r), device=dev)))
o = torch.empty(10, device=dev)
attempt("reduce_scatter_tensor", lambda: dist.reduce_scatter_tensor(o, torch.arange(20, dtype=torch.float32, device=dev)))
if r == 0:
    print(res, float(t[0]), b.tolist()[:2], g.tolist()[::10], o.tolist()[:2])
dist.destroy_process_group()
