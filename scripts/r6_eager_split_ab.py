"""A/B of the Gram split behind the feature extraction (option gram_eager_split), same box, alternating.
Per setting: stage times of hog + gram per level (HIP events of the library's timers) and the wall time of hog -> gram -> synchronize."""
import sys, time
import numpy as np
from superviseddescent_amd import Context, HoGParam, ibug, synth

SETTINGS = (0, 2, 4, 8, 16)

def run(model, n_img, per, reps=6):
    IDS = ibug.RCR22_IDS if model == "rcr22" else ibug.IBUG68_IDS
    RE, LE = ibug.eye_indices(IDS)
    images, boxes, gt = synth.make_faces(n_img, seed=9100, chunk=32, workers=16)
    x_star, x0, idx = synth.make_samples(boxes, gt, IDS, n_perturb=per - 1, seed=9101)
    hps = [HoGParam(*p) for p in ibug.SHIPPED_HOG_PARAMS]
    ctxs = {}
    for eager in SETTINGS:
        c = Context(0)
        c.set_option("gram_eager_split", eager)
        c.set_model_geometry(len(IDS), RE, LE, hps)
        c.upload_images(images); c.set_sample_image_index(idx); c.set_x(x0); c.set_targets(x_star)
        c.enable_timing(True)
        ctxs[eager] = c
    for lvl in (0, 3):
        for eager in SETTINGS:        # warm
            c = ctxs[eager]; c.hog_features(lvl); c.gram_rhs(lvl); c.synchronize(); c.get_timing(reset=True)
        rows = {e: [] for e in SETTINGS}
        for r in range(reps):
            for eager in SETTINGS:
                c = ctxs[eager]
                c.synchronize(); t0 = time.perf_counter()
                c.hog_features(lvl); c.gram_rhs(lvl); c.synchronize()
                wall = (time.perf_counter() - t0) * 1e3
                t = c.get_timing(reset=True)
                rows[eager].append((t["hog"][0], t["gram"][0], wall))
        for eager in SETTINGS:
            a = np.array(rows[eager])
            print(f"{model} rows={x0.shape[0]} level={lvl} eager={eager}: hog {np.median(a[:,0]):7.3f}  gram {np.median(a[:,1]):8.3f}  hog+gram wall {np.median(a[:,2]):8.3f} ms   (min wall {a[:,2].min():.3f})", flush=True)

if __name__ == "__main__":
    run("rcr22", 2000, 50)
    if len(sys.argv) > 1 and sys.argv[1] == "all":
        run("rcr68", 2000, 50)
