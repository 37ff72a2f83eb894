"""Is the Gram product bound by the chip's power budget?  The same launch on the real feature rows, on rows of zeros (matrix instructions on
zeros switch far fewer gates) and on rows of one constant: stage time (library HIP events, median of 5 after a warm-up) and the clock rocm-smi
reports right behind it.  Same bytes, same instructions in all three."""
import subprocess, sys
import numpy as np
import torch
from superviseddescent_amd import Context, HoGParam, ibug, synth


def span(ptr, count):
    class Span:
        __cuda_array_interface__ = {"shape": (count,), "typestr": "<f4", "data": (ptr, False), "version": 3, "strides": None}
    return torch.as_tensor(Span(), device="cuda:0")


def sclk():
    try:
        out = subprocess.run(["rocm-smi", "--showclocks"], capture_output=True, text=True, timeout=20).stdout
        for line in out.splitlines():
            if "sclk" in line:
                return line.split("(")[-1].split(")")[0]
    except Exception:
        pass
    return "?"


ids = ibug.RCR22_IDS
RE, LE = ibug.eye_indices(ids)
images, boxes, gt = synth.make_faces(2000, seed=9100, chunk=32, workers=16)
x_star, x0, idx = synth.make_samples(boxes, gt, ids, n_perturb=49, seed=9101)
c = Context(0)
c.set_model_geometry(len(ids), RE, LE, [HoGParam(*ibug.SHIPPED_HOG_PARAMS[0])])
c.upload_images(images); c.set_sample_image_index(idx); c.set_x(x0); c.set_targets(x_star)
c.enable_timing(True)
c.hog_features(0)
p, ld, n = c.features_device_ptr()
A = span(p, n * ld).view(n, ld)
for label, fill in (("real feature rows", None), ("zeros", 0.0), ("constant 0.25", 0.25), ("uniform random in [0, 0.4)", "rand")):
    if fill == "rand":
        A.uniform_(0.0, 0.4)
    elif fill is not None:
        A.fill_(fill)
    torch.cuda.synchronize()
    ts = []
    for r in range(8):
        c.get_timing(reset=True)
        c.gram_rhs(0); c.synchronize()
        ts.append(c.get_timing(reset=True)["gram"][0])
    clk = sclk()
    print(f"{label:32s} Gram stage {np.median(ts[3:]):7.3f} ms   (runs: {' '.join('%.2f' % t for t in ts)})   sclk right behind: {clk}", flush=True)
