"""A/B of the Gram launch with the tiles of its last partial round cut along the rows (option gram_ksplit), same box, alternating:
stage time of sdm_gram_rhs (library HIP events, median of 6) and the distance of the two Gram matrices from each other / from float64
on a sample of columns."""
import sys
import numpy as np
import torch
from superviseddescent_amd import Context, HoGParam, ibug, synth, parallel


def span(ptr, count):
    class Span:
        __cuda_array_interface__ = {"shape": (count,), "typestr": "<f4", "data": (ptr, False), "version": 3, "strides": None}
    return torch.as_tensor(Span(), device="cuda:0")


def run(tag, ids, hp, n_img, per, reps=6):
    RE, LE = ibug.eye_indices(ids)
    images, boxes, gt = synth.make_faces(n_img, seed=9100, chunk=32, workers=16)
    x_star, x0, idx = synth.make_samples(boxes, gt, ids, n_perturb=per - 1, seed=9101)
    c = Context(0)
    c.set_model_geometry(len(ids), RE, LE, [hp])
    c.upload_images(images); c.set_sample_image_index(idx); c.set_x(x0); c.set_targets(x_star)
    c.enable_timing(True)
    c.hog_features(0)
    F = c.feature_dim(0)
    Fp = -(-F // 128) * 128
    G = {}
    times = {0: [], 1: []}
    for r in range(reps + 1):
        for on in (0, 1):
            c.set_option("gram_ksplit", on)
            c.get_timing(reset=True)
            c.gram_rhs(0); c.synchronize()
            t = c.get_timing(reset=True)["gram"][0]
            if r > 0:
                times[on].append(t)
            if r == reps:
                ptr, count = c.gram_device_ptr()
                G[on] = span(ptr, count).view(Fp, -1).clone(); torch.cuda.synchronize()
    ncols = G[0].shape[1]
    T = ncols // 128
    up = (torch.arange(Fp, device="cuda:0")[:, None] // 128) <= (torch.arange(ncols, device="cuda:0")[None, :] // 128)
    d = ((G[1] - G[0])[up].double().norm() / G[0][up].double().norm()).item()
    # float64 reference on the last 256 columns (the cut tiles are the last in dispatch order: high tile rows / columns) and the first 128
    p, ld, n = c.features_device_ptr()
    A = span(p, n * ld).view(n, ld)
    errs = []
    for c0 in (0, Fp - 256):
        cols = slice(c0, c0 + 256)
        ref = torch.zeros((Fp, 256), dtype=torch.float64, device="cuda:0")
        for r0 in range(0, n, 20000):
            a = A[r0:r0 + 20000, :Fp].double()
            ref += a.T @ a[:, cols]
        m = up[:, cols]
        for on in (0, 1):
            errs.append(((G[on][:, cols].double() - ref)[m].norm() / ref[m].norm()).item())
    print(f"{tag}: rows {n} F {F} tiles {sum(T - 2 * i for i in range((T + 1) // 2))}   gram stage off {np.median(times[0]):.3f} ms  on {np.median(times[1]):.3f} ms"
          f"   |G_on - G_off| / |G| = {d:.2e}   vs float64 (first 256 cols off/on, last 256 cols off/on): " + " ".join(f"{e:.2e}" for e in errs), flush=True)
    c.close()


if __name__ == "__main__":
    run("RCR-22 level 0", ibug.RCR22_IDS, HoGParam(*ibug.SHIPPED_HOG_PARAMS[0]), 2000, 50)
    run("config 3 (F = 17 051, 10 000 rows)", ibug.RCR22_IDS, HoGParam(1, 5, 11, 9, 1.0), 1000, 10)
    if len(sys.argv) > 1:
        run("RCR-68 level 0", ibug.IBUG68_IDS, HoGParam(*ibug.SHIPPED_HOG_PARAMS[0]), 2000, 50)
