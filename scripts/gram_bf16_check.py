"""Accuracy and time of the Gram launch, f32 matrix-core kernel against the bf16 x 3 kernel (SDM_GRAM_BF16X3=1), on real HOG features:
relative Frobenius distance of the upper triangle {A^T A, A^T b} from a float64 product, largest entry error relative to the
largest entry, and the launch time.   python scripts/gram_bf16_check.py [rows]"""
import ctypes, json, os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

def child(rows):
    import torch
    from superviseddescent_amd import Context, HoGParam, ibug, synth
    ids = ibug.RCR22_IDS; re, le = ibug.eye_indices(ids)
    images, boxes, gt = synth.make_faces(256, seed=3)
    per = -(-rows // 256)
    x_star, x0, idx = synth.make_samples(boxes, gt, ids, n_perturb=per - 1, seed=4)
    x_star, x0, idx = x_star[:rows], x0[:rows], idx[:rows]
    ctx = Context(0)
    ctx.set_model_geometry(len(ids), re, le, [HoGParam(1, 3, 12, 4, 0.9)])      # F = 3169
    ctx.upload_images(images); ctx.set_sample_image_index(idx); ctx.set_x(x0); ctx.set_targets(x_star)
    A = ctx.hog_features(0, fetch=True).astype(np.float64)
    ctx.enable_timing(True)
    best = 1e9
    for _ in range(3):
        ctx.get_timing(reset=True); ctx.gram_rhs(0); ctx.synchronize()
        best = min(best, ctx.get_timing(reset=True)["gram"][0])
    ptr, count = ctx.gram_device_ptr()
    F = A.shape[1]
    ncols = -(-F // 128) * 128 + 128
    class Span:
        __cuda_array_interface__ = {"shape": (count,), "typestr": "<f4", "data": (ptr, False), "version": 3, "strides": None}
    G = torch.as_tensor(Span(), device="cuda:0").cpu().numpy().reshape(-1, ncols).astype(np.float64)
    ref = A.T @ A
    iu = np.triu_indices(F)
    d = G[:F, :F][iu] - ref[iu]
    print(json.dumps({"bf16x3": os.environ.get("SDM_GRAM_BF16X3", "0"), "rows": rows, "features": F, "gram_ms": best,
                      "rel_fro_vs_f64": float(np.linalg.norm(d) / np.linalg.norm(ref[iu])),
                      "max_abs_err_over_max_entry": float(np.abs(d).max() / np.abs(ref).max()),
                      "min_eig_shift_ok": bool(np.isfinite(G[:F, :F][iu]).all())}), flush=True)

if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        child(int(sys.argv[2]))
    else:
        rows = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
        for m in ("0", "1"):
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(rows)], env=dict(os.environ, SDM_GRAM_BF16X3=m),
                               capture_output=True, text=True, timeout=900)
            print(r.stdout.strip() or r.stderr[-2000:], flush=True)
