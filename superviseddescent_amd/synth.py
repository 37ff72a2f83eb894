"""Deterministic synthetic faces for benchmarks and parity tests.

No dataset can be downloaded here, so the workload is generated: 256x256 single-channel u8 images
made of a smooth random background, per-pixel noise and a small blob at each of the 68 ground-truth
ibug landmark positions (so that HOG patches around the landmarks carry signal and a trained cascade
actually converges).  Ground truth and initialisations follow the reference trainer's recipe:
``x* = align_mean(mean, box) + N(0, 1.5 px)``, ``x0 = align_mean(mean, perturb(box, t~N(0,.04),
t~N(0,.04), s~N(1,.04)))`` (apps/rcr/rcr-train.cpp:130-146, 387-397, 421-431; include/rcr/model.hpp:64-76).

The same bytes are fed to the GPU engine and to the CPU oracle.
"""
from __future__ import annotations

import numpy as np

from . import ibug

SEED = 0x5D2015
IMAGE_SIZE = 256


def align_mean(mean: np.ndarray, box, scaling_x=1.0, scaling_y=1.0, translation_x=0.0,
               translation_y=0.0) -> np.ndarray:
    """rcr::align_mean (include/rcr/model.hpp:64-76), evaluated in float32.  box = (x, y, w, h)."""
    mean = np.asarray(mean, np.float32).reshape(-1)
    L = mean.size // 2
    f = np.float32
    out = np.empty_like(mean)
    out[:L] = (mean[:L] * f(scaling_x) + f(0.5) + f(translation_x)) * f(box[2]) + f(box[0])
    out[L:] = (mean[L:] * f(scaling_y) + f(0.5) + f(translation_y)) * f(box[3]) + f(box[1])
    return out


def perturb(box, tx: float, ty: float, s: float = 1.0):
    """perturb() of apps/rcr/rcr-train.cpp:130-146: float arithmetic, truncated into an int Rect."""
    f = np.float32
    x, y, w, h = (int(v) for v in box)
    pw, ph = f(w) * f(s), f(h) * f(s)
    nx = f(x) + (f(w) - pw) / f(2.0) + f(tx) * f(w)
    ny = f(y) + (f(h) - ph) / f(2.0) + f(ty) * f(h)
    return (int(nx), int(ny), int(pw), int(ph))


def _face_chunk(rng, n: int, size: int):
    """One chunk of make_faces: (images [n,size,size] u8, boxes [n,4], gt68 [n,136]) drawn from ``rng``."""
    coords = np.arange(size, dtype=np.float32)
    wh = rng.integers(160, 209, size=n)
    bx = (rng.random(n) * (size - wh)).astype(np.int32)
    by = (rng.random(n) * (size - wh)).astype(np.int32)
    boxes = np.stack([bx, by, wh, wh], 1).astype(np.int32)
    # smooth background: 6 Gaussians (sigma 12..40 px), amplitude +-48
    gcx = rng.random((n, 6)).astype(np.float32) * size
    gcy = rng.random((n, 6)).astype(np.float32) * size
    gsig = (12 + 28 * rng.random((n, 6))).astype(np.float32)
    gamp = (48 * (2 * rng.random((n, 6)) - 1)).astype(np.float32)
    ex = np.exp(-0.5 * ((coords[None, None, :] - gcx[:, :, None]) / gsig[:, :, None]) ** 2)
    ey = np.exp(-0.5 * ((coords[None, None, :] - gcy[:, :, None]) / gsig[:, :, None]) ** 2)
    field = np.matmul((ey * gamp[:, :, None]).transpose(0, 2, 1), ex)          # [n, y, x]
    # landmark blobs at the ground-truth shape (rigid mean + 1.5 px jitter)
    gt68 = np.empty((n, 136), np.float32)
    for i in range(n):
        gt68[i] = align_mean(ibug.MEAN_IBUG_LFPW_68, boxes[i])
    gt68 += (1.5 * rng.standard_normal((n, 136))).astype(np.float32)
    lx, ly = gt68[:, :68], gt68[:, 68:]
    lamp = np.where(np.arange(68) % 2 == 0, 70.0, -70.0).astype(np.float32)
    bxk = np.exp(-0.5 * ((coords[None, None, :] - lx[:, :, None]) / 3.0) ** 2).astype(np.float32)
    byk = np.exp(-0.5 * ((coords[None, None, :] - ly[:, :, None]) / 3.0) ** 2).astype(np.float32)
    field += np.matmul((byk * lamp[None, :, None]).transpose(0, 2, 1), bxk)
    noise = (24 * (2 * rng.random((n, size, size), dtype=np.float32) - 1))
    return np.clip(128 + field + noise, 0, 255).astype(np.uint8), boxes, gt68


def make_faces(n_images: int, seed: int = SEED, size: int = IMAGE_SIZE, chunk: int = 256, workers: int = 0):
    """Returns (images uint8 [n,size,size], boxes int32 [n,4] (x,y,w,h), gt68 float32 [n,136]).

    ``workers <= 1``: one random stream, chunk after chunk (the stream every fixture and test of this repo uses).
    ``workers > 1``: the chunks are drawn from independent streams ``default_rng([seed, chunk_index])`` in a pool of
    forked worker processes -- a different (equally deterministic) data set, meant for the large training sets of bench.py."""
    images = np.empty((n_images, size, size), np.uint8)
    boxes = np.empty((n_images, 4), np.int32)
    gt68 = np.empty((n_images, 136), np.float32)
    starts = list(range(0, n_images, chunk))
    if workers <= 1:
        rng = np.random.default_rng(seed)
        for c0 in starts:
            n = min(chunk, n_images - c0)
            images[c0:c0 + n], boxes[c0:c0 + n], gt68[c0:c0 + n] = _face_chunk(rng, n, size)
        return images, boxes, gt68
    import multiprocessing
    from concurrent.futures import ProcessPoolExecutor

    jobs = [(seed, ci, min(chunk, n_images - c0), size) for ci, c0 in enumerate(starts)]
    # fork: the workers only run numpy.  Call this BEFORE the process initialises HIP / torch.cuda: forking an
    # initialised ROCm process leaves a later torch.cuda initialisation without devices
    with ProcessPoolExecutor(max_workers=workers, mp_context=multiprocessing.get_context("fork")) as pool:
        for (_, ci, n, _), (im, bx, gt) in zip(jobs, pool.map(_face_chunk_job, jobs)):
            c0 = starts[ci]
            images[c0:c0 + n], boxes[c0:c0 + n], gt68[c0:c0 + n] = im, bx, gt
    return images, boxes, gt68


def _face_chunk_job(job):
    seed, ci, n, size = job
    return _face_chunk(np.random.default_rng([seed, ci]), n, size)


def make_samples(boxes: np.ndarray, gt68: np.ndarray, landmark_ids, n_perturb: int = 0,
                 seed: int = SEED + 1):
    """Rows of the training/test matrices.  For every image the unperturbed box, then ``n_perturb``
    perturbed boxes (rcr-train.cpp:421-431).  Returns (x_star [N,2L], x0 [N,2L], img_index [N])."""
    rng = np.random.default_rng(seed)
    pos = [ibug.IBUG68_IDS.index(i) for i in landmark_ids]
    sel = np.array(pos + [68 + p for p in pos])
    mean_sel = ibug.select_mean(landmark_ids)
    n_img = boxes.shape[0]
    per = 1 + n_perturb
    N = n_img * per
    x_star = np.repeat(gt68[:, sel], per, axis=0).astype(np.float32)
    x0 = np.empty((N, sel.size), np.float32)
    img_index = np.repeat(np.arange(n_img, dtype=np.int32), per)
    t = rng.normal(0.0, 0.04, size=(N, 2))
    s = rng.normal(1.0, 0.04, size=N)
    for r in range(N):
        box = boxes[img_index[r]]
        if n_perturb == 0 or r % per != 0:
            box = perturb(box, t[r, 0], t[r, 1], s[r])
        x0[r] = align_mean(mean_sel, box)
    return x_star, x0, img_index
