import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from superviseddescent_amd import Context, HoGParam, ibug, synth
ids = ibug.RCR22_IDS; re, le = ibug.eye_indices(ids)
params = [HoGParam(1, 5, c, 9, r) for c, r in ((11, 1.0), (10, 0.7), (8, 0.4), (6, 0.25), (6, 0.25))]
images, boxes, gt = synth.make_faces(1000, seed=1, chunk=32, workers=16)
xs, x0, idx = synth.make_samples(boxes, gt, ids, n_perturb=9, seed=2)
c = Context(0)
def T(name, fn):
    c.synchronize(); t = time.perf_counter(); r = fn(); c.synchronize(); print(f"  {name:18s} {(time.perf_counter()-t)*1e3:8.2f} ms"); return r
for rep in range(2):
    print("pass", rep)
    T("set_model_geometry", lambda: c.set_model_geometry(len(ids), re, le, params))
    T("upload_images", lambda: c.upload_images(images))
    T("set_idx", lambda: c.set_sample_image_index(idx))
    T("set_x", lambda: c.set_x(x0)); T("set_targets", lambda: c.set_targets(xs))
    for l in range(2):
        T("hog", lambda: c.hog_features(l)); T("gram", lambda: c.gram_rhs(l))
        T("solve", lambda: c.solve(l, 0, 1.0, True, xs.shape[0])); T("apply", lambda: c.apply(l))
print("repeat apply / hog only")
for _ in range(3):
    T("hog", lambda: c.hog_features(1)); T("apply", lambda: c.apply(1)); T("apply again (same feats)", lambda: (c.hog_features(1), c.apply(1)))
print("fine timing after solve")
for l in range(2):
    c.hog_features(l); c.gram_rhs(l); c.solve(l, 0, 1.0, True, xs.shape[0]); c.synchronize()
    t0 = time.perf_counter(); c.apply(l); t1 = time.perf_counter(); c.synchronize(); t2 = time.perf_counter()
    print(f"  apply call {1e3*(t1-t0):.2f} ms, sync {1e3*(t2-t1):.2f} ms")
    t0 = time.perf_counter(); c.hog_features(l); t1 = time.perf_counter(); c.synchronize(); t2 = time.perf_counter()
    print(f"  hog call {1e3*(t1-t0):.2f} ms, sync {1e3*(t2-t1):.2f} ms")
print("sync latency after solve with a trivial follow-up")
for l in range(2):
    c.hog_features(l); c.gram_rhs(l); c.solve(l, 0, 1.0, True, xs.shape[0]); c.synchronize()
    t0 = time.perf_counter(); c.set_targets(xs); t1 = time.perf_counter(); c.synchronize(); t2 = time.perf_counter()
    print(f"  set_targets (H2D + sync inside) {1e3*(t1-t0):.2f} ms, extra sync {1e3*(t2-t1):.2f} ms")
    c.hog_features(l); c.gram_rhs(l); c.solve(l, 0, 1.0, True, xs.shape[0]); c.synchronize(); time.sleep(0.05)
    t0 = time.perf_counter(); c.apply(l); c.synchronize(); t2 = time.perf_counter()
    print(f"  apply+sync after a 50 ms pause: {1e3*(t2-t0):.2f} ms")
print("bisect: what before the slow follow-up?")
import ctypes
def follow(tag):
    t0 = time.perf_counter(); c.set_targets(xs); t1 = time.perf_counter()
    print(f"  {tag:34s} follow-up set_targets {1e3*(t1-t0):.2f} ms")
c.hog_features(0); c.synchronize(); follow("after hog")
c.hog_features(0); c.gram_rhs(0); c.synchronize(); follow("after hog+gram")
c.hog_features(0); c.gram_rhs(0); c.solve(0, 0, 1.0, True, xs.shape[0], fetch=False); c.synchronize(); follow("after solve(fetch=False)")
c.hog_features(0); c.gram_rhs(0); c.solve(0, 0, 1.0, True, xs.shape[0]); c.synchronize(); follow("after solve(fetch=True)")
c.hog_features(0); c.gram_rhs(0); c.solve(0, 0, 1.0, True, xs.shape[0]); c.synchronize(); c.synchronize(); follow("after solve + 2 syncs")
