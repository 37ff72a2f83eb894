"""Host-layer behaviour on the device: the image a HogTransform reads is the image's CURRENT content
(adaptive_vlhog.hpp:109-120 reads `images[training_index]` at every call), also when the caller refills one frame buffer;
regressors the device already holds are not uploaded again; `detect` tells a face box from an initialisation by argument
form, not by `size == 4`; the sample -> image index is bounds-checked at launch."""
import numpy as np
import pytest

from superviseddescent_amd import (Context, HoGParam, HogTransform, LinearRegressor, SdmError, SupervisedDescentOptimiser,
                                   detection_model, ibug, synth)

pytestmark = pytest.mark.gpu

IDS = ibug.RCR22_IDS


def make_model(n_levels=2, seed=1):
    rng = np.random.default_rng(seed)
    params = [HoGParam(*p) for p in ibug.SHIPPED_HOG_PARAMS[:n_levels]]
    regs = []
    for _ in range(n_levels):
        r = LinearRegressor()
        r.x = (rng.standard_normal((8801, 44)) * 0.003).astype(np.float32)
        regs.append(r)
    return detection_model(SupervisedDescentOptimiser(regs), ibug.select_mean(IDS), IDS, params, ibug.RIGHT_EYE_IDS,
                           ibug.LEFT_EYE_IDS), regs


def test_detect_reads_the_current_frame_and_keeps_regressors_resident(built):
    images, boxes, _ = synth.make_faces(4, seed=808)
    model, regs = make_model()
    want = [make_model()[0].detect(images[i].copy(), boxes[i]) for i in range(3)]      # fresh model + fresh array per frame
    ctx = model.optimised_model.ctx
    uploads = []
    orig = ctx.set_regressor
    ctx.set_regressor = lambda level, R, **kw: (uploads.append(level), orig(level, R, **kw))[1]
    frame = np.empty_like(images[0])                     # ONE buffer, refilled per frame as a video loop does
    got = []
    for i in range(3):
        frame[:] = images[i]
        got.append(model.detect(frame, boxes[i]))
    for g, w in zip(got, want):
        assert np.array_equal(g, w)
    assert uploads == [0, 1]                             # uploaded for the first frame only
    regs[1].x = regs[1].x * np.float32(0.5)              # a new matrix -> that level (only) goes up again
    model.detect(frame, boxes[2])
    assert uploads == [0, 1, 1]
    with pytest.raises(ValueError):                      # a resident matrix is read-only: an in-place edit cannot go unnoticed
        regs[0].x[:] = 0.0
    with pytest.raises(ValueError):
        regs[0].x *= np.float32(2.0)
    regs[0].touch()                                      # touch() + in-place change
    regs[0].x[:] = 0.0
    model.detect(frame, boxes[2])
    assert uploads == [0, 1, 1, 0]


def test_two_optimisers_sharing_a_context_do_not_see_each_others_regressors(built):
    """the record of what the device holds lives in the Context, names the array object and is replaced by
    whoever writes the level."""
    images, boxes, _ = synth.make_faces(2, seed=814)
    model_a, regs_a = make_model(seed=1)
    ctx = model_a.optimised_model.ctx
    rng = np.random.default_rng(2)
    regs_b = []
    for _ in range(2):
        r = LinearRegressor()
        r.x = (rng.standard_normal((8801, 44)) * 0.003).astype(np.float32)
        regs_b.append(r)
    model_b = detection_model(SupervisedDescentOptimiser(regs_b, ctx=ctx), ibug.select_mean(IDS), IDS, model_a.hog_params,
                              ibug.RIGHT_EYE_IDS, ibug.LEFT_EYE_IDS)
    want_a = make_model(seed=1)[0].detect(images[0], boxes[0])
    a1 = model_a.detect(images[0], boxes[0])
    b1 = model_b.detect(images[0], boxes[0])             # same context, same geometry, other regressors
    a2 = model_a.detect(images[0], boxes[0])             # must not run with B's coefficients
    assert np.array_equal(a1, want_a) and np.array_equal(a2, want_a)
    assert not np.array_equal(b1, a1)
    # swapping in NEW regressor objects (whose id() may reuse a freed one's) is seen as well
    opt = model_a.optimised_model
    for l in range(2):
        fresh = LinearRegressor()
        fresh.x = regs_b[l].x.copy()
        opt.regressors[l] = fresh
    assert np.array_equal(model_a.detect(images[0], boxes[0]), b1)
    # a direct Context.set_regressor by a third party invalidates the record too
    ctx.set_regressor(0, np.zeros((8801, 44), np.float32))
    assert np.array_equal(model_a.detect(images[0], boxes[0]), b1)


def test_resident_images_are_an_explicit_opt_in(built):
    images, boxes, gt = synth.make_faces(6, seed=809)
    _, x0, _ = synth.make_samples(boxes, gt, IDS, 0, seed=810)
    model, _ = make_model()
    sdo = model.optimised_model
    stack = images.copy()
    hog = HogTransform(stack, model.hog_params, IDS, ibug.RIGHT_EYE_IDS, ibug.LEFT_EYE_IDS, images_resident=True)
    a = sdo.test(x0, None, hog)
    stack[:] = stack[::-1].copy()                        # the caller broke its promise: the device still holds the old pixels
    assert np.array_equal(sdo.test(x0, None, hog), a)
    hog2 = HogTransform(stack, model.hog_params, IDS, ibug.RIGHT_EYE_IDS, ibug.LEFT_EYE_IDS)      # default: always current
    b = sdo.test(x0, None, hog2)
    assert not np.array_equal(a, b)
    stack[:] = stack[::-1].copy()
    assert np.array_equal(sdo.test(x0, None, hog2), a)


def test_detect_argument_forms_with_a_two_landmark_model(built):
    """2L = 4: an initialisation row has as many numbers as a face box."""
    images, boxes, _ = synth.make_faces(2, seed=811)
    ids = ["37", "46"]
    rng = np.random.default_rng(3)
    params = [HoGParam(1, 5, 6, 4, 0.6)]
    reg = LinearRegressor()
    reg.x = (rng.standard_normal((2 * 25 * 16 + 1, 4)) * 0.01).astype(np.float32)
    mean = ibug.select_mean(ids)
    model = detection_model(SupervisedDescentOptimiser([reg]), mean, ids, params, ["37"], ["46"])
    box = tuple(int(v) for v in boxes[0])
    from_box = model.detect(images[0], box)
    init = synth.align_mean(mean, box)
    assert np.array_equal(model.detect(images[0], initialisation=init), from_box)
    assert np.array_equal(model.detect(images[0], init.reshape(1, 4)), from_box)          # 2-D second argument = initialisation
    other = model.detect(images[0], np.array(box, np.float32))                             # 1-D four numbers = face box
    assert np.array_equal(other, from_box)
    with pytest.raises(ValueError):
        model.detect(images[0])
    with pytest.raises(ValueError):
        model.detect(images[0], box, initialisation=init)


def test_sample_image_index_is_checked_at_launch(built):
    images, boxes, gt = synth.make_faces(4, seed=812)
    _, x0, _ = synth.make_samples(boxes, gt, IDS, 1, seed=813)                             # 8 rows over 4 images
    re, le = ibug.eye_indices(IDS)
    ctx = Context(0)
    ctx.set_model_geometry(len(IDS), re, le, [HoGParam(*ibug.SHIPPED_HOG_PARAMS[3])])
    ctx.upload_images(images)
    ctx.set_x(x0)
    ctx.set_sample_image_index(np.array([0, 1, 2], np.int32))                              # shorter than the 8 rows
    with pytest.raises(SdmError, match="shorter than the sample count"):
        ctx.hog_features(0)
    ctx.set_sample_image_index(np.repeat(np.arange(4), 2).astype(np.int32))
    ctx.hog_features(0)
    ctx.upload_images(images[:2])                                                          # fewer images than the index refers to
    with pytest.raises(SdmError, match="beyond the current image set"):
        ctx.hog_features(0)
    ctx.set_sample_image_index(None)
    with pytest.raises(SdmError, match="more samples than images"):
        ctx.hog_features(0)
    # the optimiser checks the length of HogTransform.img_index before anything reaches the device
    model, _ = make_model(1)
    hog = HogTransform(images, model.hog_params, IDS, ibug.RIGHT_EYE_IDS, ibug.LEFT_EYE_IDS, img_index=np.array([0, 1], np.int32))
    with pytest.raises(ValueError, match="one image number per sample row"):
        model.optimised_model.test(x0, None, hog)
    ctx.close()
