"""Round 5 probe: stage times of an RCR-68 (or RCR-22) training level at the bench's shape (rows x F), second of two passes.
    python scripts/r5_rcr68_train_probe.py [rows] [68|22]"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from superviseddescent_amd import ibug, synth
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
which = sys.argv[2] if len(sys.argv) > 2 else "68"
ids = ibug.IBUG68_IDS if which == "68" else ibug.RCR22_IDS
timg, tbox, tgt = synth.make_faces(rows // 10, seed=synth.SEED + 1000, chunk=32, workers=16)
txs, tx0, tidx = synth.make_samples(tbox, tgt, ids, n_perturb=9, seed=synth.SEED + 2000)
import torch
from superviseddescent_amd import HoGParam, HogTransform, LinearRegressor, Regulariser, SupervisedDescentOptimiser
params = [HoGParam(*p) for p in ibug.SHIPPED_HOG_PARAMS]
stream = torch.cuda.current_stream().cuda_stream if os.environ.get("PROBE_TORCH_STREAM", "1") == "1" else None
sdo = SupervisedDescentOptimiser([LinearRegressor(Regulariser(1, 1.5, False)) for _ in params], device=0, stream=stream)
hog = HogTransform(timg, params, ids, ibug.RIGHT_EYE_IDS, ibug.LEFT_EYE_IDS, tidx, images_resident=True)
for rep in range(2):
    sdo.ctx.enable_timing(True); sdo.ctx.get_timing(reset=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    sdo.train(txs, tx0, None, hog)
    torch.cuda.synchronize(); wall = time.perf_counter() - t0
    tm = sdo.ctx.get_timing(reset=True)
print(json.dumps({"rows": rows, "landmarks": which, "chain_v1": os.environ.get("SDM_SOLVE_CHAIN_V1", "0"), "torch_stream": os.environ.get("PROBE_TORCH_STREAM", "1"),
                  "sec_per_cascade": wall / 4, "stage_ms_per_level": {k: round(v[0] / 4, 3) for k, v in tm.items() if v[1] > 0}}))
