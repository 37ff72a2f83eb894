"""Round 6: the four-wave Gram kernel against the eight-wave kernel of rounds 3-5 (SDM_GRAM_KERNEL = w8p | w4check | unset).
  1. accuracy + bits on real HOG features (F = 3 169, 20 000 rows): distance from a float64 product; w4check must equal w8p bit for bit
  2. time of the Gram stage at RCR-22 / 100 000 rows and (optionally) RCR-68 / 100 000 rows
    python scripts/r6_gram_ab.py [--big]
(The w8p / w4check legs need scripts/experiments/gram_w8p_kernels.patch applied in reverse: the shipped library has the four-wave kernel only;
the recorded run is profiles/r06_gram_ab.txt.)  Every kernel runs in its own process under a timeout (a hand-placed instruction stream that waits wrongly hangs)."""
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "gpurun_out")


def gram_matrix(ctx, F):
    import torch
    ptr, count = ctx.gram_device_ptr()
    ncols = -(-F // 128) * 128 + 128

    class Span:
        __cuda_array_interface__ = {"shape": (count,), "typestr": "<f4", "data": (ptr, False), "version": 3, "strides": None}
    return torch.as_tensor(Span(), device="cuda:0").cpu().numpy().reshape(-1, ncols)


def child_accuracy(tag, rows):
    from superviseddescent_amd import Context, HoGParam, ibug, synth
    ids = ibug.RCR22_IDS
    re, le = ibug.eye_indices(ids)
    images, boxes, gt = synth.make_faces(256, seed=3)
    per = -(-rows // 256)
    x_star, x0, idx = synth.make_samples(boxes, gt, ids, n_perturb=per - 1, seed=4)
    x_star, x0, idx = x_star[:rows], x0[:rows], idx[:rows]
    ctx = Context(0)
    ctx.set_model_geometry(len(ids), re, le, [HoGParam(1, 3, 12, 4, 0.9)])      # F = 3169
    ctx.upload_images(images)
    ctx.set_sample_image_index(idx)
    ctx.set_x(x0)
    ctx.set_targets(x_star)
    A = ctx.hog_features(0, fetch=True).astype(np.float64)
    ctx.gram_rhs(0)
    ctx.synchronize()
    F = A.shape[1]
    G = gram_matrix(ctx, F)
    ref = A.T @ A
    T = -(-F // 128)
    # the valid part: 128 x 128 tiles with tile row <= tile column
    mask = np.zeros(G.shape, bool)
    for ti in range(T):
        mask[ti * 128:(ti + 1) * 128, ti * 128:] = True
    mask = mask[:F, :F]
    d = (G[:F, :F].astype(np.float64) - ref)[mask]
    np.save(os.path.join(OUT, f"r6_gram_{tag}.npy"), G)
    print(json.dumps({"kernel": tag, "rows": rows, "features": F, "rel_fro_vs_f64": float(np.linalg.norm(d) / np.linalg.norm(ref[mask])),
                      "max_abs_err_over_max_entry": float(np.abs(d).max() / np.abs(ref).max()), "finite": bool(np.isfinite(G).all())}), flush=True)


def child_time(tag, rows, hp, n_ids):
    from superviseddescent_amd import Context, HoGParam, ibug, synth
    ids = ibug.RCR22_IDS if n_ids == 22 else ibug.IBUG68_IDS
    re, le = ibug.eye_indices(ids)
    images, boxes, gt = synth.make_faces(256, seed=1)
    per = -(-rows // 256)
    x_star, x0, idx = synth.make_samples(boxes, gt, ids, n_perturb=per - 1, seed=2)
    x_star, x0, idx = x_star[:rows], x0[:rows], idx[:rows]
    ctx = Context(0)
    ctx.set_model_geometry(len(ids), re, le, [HoGParam(*hp)])
    ctx.upload_images(images)
    ctx.set_sample_image_index(idx)
    ctx.set_x(x0)
    ctx.set_targets(x_star)
    ctx.enable_timing(True)
    ctx.hog_features(0)
    times = []
    for _ in range(5):
        ctx.get_timing(reset=True)
        ctx.gram_rhs(0)
        ctx.synchronize()
        times.append(ctx.get_timing(reset=True)["gram"][0])
    F = len(ids) * hp[1] ** 2 * (3 * hp[3] + 4) + 1
    T = -(-F // 128)
    M = 2 * len(ids)
    Tr = -(-M // 128)
    executed = 2.0 * rows * 128 * 128 * (T * (T + 1) / 2 + T * Tr) * 3        # three float16 products per product
    best = min(times)
    print(json.dumps({"kernel": tag, "rows": rows, "F": F, "gram_ms": best, "all_ms": [round(t, 3) for t in times],
                      "frac_of_2500TF": executed / (best * 1e-3) / 2.5e15}), flush=True)


def run(tag, args, timeout):
    env = dict(os.environ)
    if tag != "w4":
        env["SDM_GRAM_KERNEL"] = tag
    else:
        env.pop("SDM_GRAM_KERNEL", None)
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", tag] + [str(a) for a in args], env=env, capture_output=True, text=True, timeout=timeout)
        print(r.stdout.strip() or ("FAILED rc=%d: " % r.returncode + r.stderr[-1500:]), flush=True)
        return r.returncode == 0
    except subprocess.TimeoutExpired:
        print(json.dumps({"kernel": tag, "args": args, "error": "timeout"}), flush=True)
        return False


def main():
    os.makedirs(OUT, exist_ok=True)
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        tag, what = sys.argv[2], sys.argv[3]
        if what == "acc":
            child_accuracy(tag, int(sys.argv[4]))
        else:
            hp = tuple(int(v) for v in sys.argv[6].split(",")[:4]) + (float(sys.argv[6].split(",")[4]),)
            child_time(tag, int(sys.argv[4]), hp, int(sys.argv[5]))
        return
    ok = {}
    for tag in ("w8p", "w4check", "w4"):
        ok[tag] = run(tag, ["acc", 20000], 240)
    try:
        a = np.load(os.path.join(OUT, "r6_gram_w8p.npy"))
        for tag in ("w4check", "w4"):
            b = np.load(os.path.join(OUT, f"r6_gram_{tag}.npy"))
            T = a.shape[0] // 128
            same = True
            worst = 0.0
            for ti in range(T):
                blk_a, blk_b = a[ti * 128:(ti + 1) * 128, ti * 128:], b[ti * 128:(ti + 1) * 128, ti * 128:]
                same = same and np.array_equal(blk_a, blk_b)
                worst = max(worst, float(np.abs(blk_a - blk_b).max()))
            print(json.dumps({"compare": f"w8p vs {tag}", "bit_identical_valid_tiles": bool(same), "max_abs_diff": worst, "max_entry": float(np.abs(a).max())}), flush=True)
    except Exception as e:      # noqa: BLE001
        print("compare failed:", e, flush=True)
    for f in os.listdir(OUT):
        if f.startswith("r6_gram_") and f.endswith(".npy"):
            os.remove(os.path.join(OUT, f))
    if not ok.get("w4"):
        return
    for tag in ("w8p", "w4"):
        run(tag, ["time", 100000, 22, "1,5,11,4,1.0"], 400)
    if "--big" in sys.argv:
        for tag in ("w8p", "w4"):
            run(tag, ["time", 100000, 68, "1,5,11,4,1.0"], 600)


if __name__ == "__main__":
    main()
