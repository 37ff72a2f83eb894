import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from superviseddescent_amd import Context, HoGParam
G = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "hog_ref_vectors.npz"))
params = [HoGParam(int(v), int(C), int(c), int(O), float(r)) for (v, C, c, O), r in zip(G["tr_params"], G["tr_rel"])]
ctx = Context(0)
ctx.set_model_geometry(5, [int(G["tr_eyes"][0])], [int(G["tr_eyes"][1])], params)
ctx.upload_images(G["tr_images"]); ctx.set_sample_image_index(None); ctx.set_x(G["tr_x"])
for li in range(len(params)):
    want = G[f"tr_feat_{li}"]
    P = (want.shape[1] - 1) // 5
    for mode in (1, 2):
        ctx.set_hog_mode(mode)
        got = ctx.hog_features(li, fetch=True)
        d = np.abs(got - want)
        print("level", li, "mode", mode, "P", P, "max", d.max(), "nan", np.isnan(got).sum())
        if d.max() > 1e-6:
            for s in range(3):
                for l in range(5):
                    seg = d[s, l * P:(l + 1) * P]
                    if seg.max() > 1e-6:
                        bad = np.nonzero(seg > 1e-6)[0]
                        print("   sample", s, "landmark", l, "bad", bad.size, "first", bad[:12], "got", got[s, l * P + bad[:4]], "want", want[s, l * P + bad[:4]])
