"""CPU checks of the lane-packing plan of the HOG launch (superviseddescent_amd/csrc/sdm_kernels.h: HogPlanDev, built by
sdm_hog_plan_build): every interior pixel column of every patch contributes exactly once, always with its true left and
right neighbour in the adjacent lanes (the gradient of hog.c:635-636 is taken across lanes), and the fold weights are the
bilinear cell weights of hog.c:697-704 for the cell columns of the right patch."""
import numpy as np
import pytest

from superviseddescent_amd.engine import hog_plan

GEOMS = [(5, cell, 4, L) for cell in (2, 4, 6, 8, 10, 11, 12) for L in (1, 2, 5, 22, 68)]


def cell_weights(col, cell, C):
    """hog.c:697-704 for pixel coordinate `col`: (cell index floor(hx), weight of that cell, weight of the next one)."""
    hx = np.float32((col + 0.5) / cell - 0.5)
    b = int(np.floor(hx))
    w2 = np.float32(hx - np.float32(b))
    w1 = np.float32(1.0 - float(w2))
    return b, w1, w2


def unpack(desc):
    return dict(slot=int(desc & 0xff), col=int((desc >> 8) & 0xff), active=int((desc >> 16) & 1),
                in_use=int((desc >> 17) & 1), seg=int((desc >> 20) & 3))


@pytest.mark.parametrize("C,cell,O,L", GEOMS)
def test_plan_covers_every_column_once(built, C, cell, O, L):
    plan = hog_plan(C, cell, O, L)
    assert plan is not None
    S = C * cell
    G, P, nm, Gt, Pt = (plan[k] for k in ("G", "P", "n_main", "Gt", "Pt"))
    assert nm * G + Gt == L and (Gt == 0) == (Pt == 0) and 0 <= Gt < G
    for first, npass, npatch in ((0, P, G), (P, Pt, Gt)):
        if npatch == 0:
            continue
        seen = np.zeros((npatch, S), int)
        done_all = []
        for t in range(first, first + npass):
            lanes = [unpack(d) for d in plan["lane_tab"][t]]
            info = plan["pass_info"][t]
            seg_slots = {}
            for x, a in enumerate(lanes):
                assert a["slot"] < npatch and a["col"] < S          # also for idle lanes: they load a landmark of the group
                if not a["in_use"]:
                    assert not a["active"]
                    continue
                assert a["seg"] < 3
                seg_slots.setdefault(a["seg"], a["slot"])
                assert seg_slots[a["seg"]] == a["slot"]
                if a["active"]:
                    assert 1 <= a["col"] <= S - 2
                    seen[a["slot"], a["col"]] += 1
                    # the DPP neighbours are the true neighbours of the column
                    for dx in (-1, 1):
                        nb = lanes[x + dx]
                        assert nb["in_use"] and nb["slot"] == a["slot"] and nb["col"] == a["col"] + dx, (t, x, dx)
            # segments are consecutive patch slots -> distinct histogram slots (slot % 3)
            slots = [seg_slots[k] for k in sorted(seg_slots)]
            assert slots == list(range(slots[0], slots[0] + len(slots)))
            for k in range(3):
                assert info[k] == seg_slots.get(k, -1)
            dfirst, dcount = int(info[3]) & 0xff, (int(info[3]) >> 8) & 0xff
            nkp, first_bits = (int(info[3]) >> 16) & 0xff, (int(info[3]) >> 24) & 7
            in_use = sum(a["in_use"] for a in lanes)
            assert nkp == -(-in_use // 8) and all(a["in_use"] == (x < in_use) for x, a in enumerate(lanes))
            for k in sorted(seg_slots):      # a segment stores (instead of adds) exactly when its patch's column 0 is in this pass
                starts = any(a["in_use"] and a["seg"] == k and a["col"] == 0 for a in lanes)
                assert bool((first_bits >> k) & 1) == starts
            done_here = [a["slot"] for a in lanes if a["in_use"] and a["col"] == S - 1]
            assert done_here == list(range(dfirst, dfirst + dcount))
            done_all += done_here
            # a patch completed here has no column in a later pass
            for t2 in range(t + 1, first + npass):
                later = {unpack(d)["slot"] for d in plan["lane_tab"][t2] if (int(d) >> 17) & 1}
                assert not (later & set(done_here))
        assert done_all == list(range(npatch))
        assert (seen[:, 1:S - 1] == 1).all() and (seen[:, 0] == 0).all() and (seen[:, S - 1] == 0).all()
    # packing never needs more passes than patches
    assert P <= G and Pt <= max(Gt, 0)


@pytest.mark.parametrize("C,cell,O,L", [(5, 11, 4, 22), (5, 10, 4, 22), (5, 8, 4, 22), (5, 6, 4, 68), (5, 10, 4, 68)])
def test_plan_fold_weights(built, C, cell, O, L):
    plan = hog_plan(C, cell, O, L)
    S = C * cell
    rng = np.random.default_rng(7)
    for t in range(plan["P"] + plan["Pt"]):
        lanes = [unpack(d) for d in plan["lane_tab"][t]]
        wb = plan["wb"][t]                              # [lane][16]: entry ks of lane l = W[4 ks + (l >> 4)][l & 15]
        W = np.zeros((64, 16), np.float32)
        for l in range(64):
            for ks in range(16):
                W[4 * ks + (l >> 4), l & 15] = wb[l, ks]
        for x, a in enumerate(lanes):
            want = np.zeros(16, np.float32)
            if a["active"]:
                b, w1, w2 = cell_weights(a["col"], cell, C)
                if b >= 0:
                    want[a["seg"] * C + b] = w1
                if b + 1 <= C - 1:
                    want[a["seg"] * C + b + 1] = w2
            assert (W[x] == want).all(), (t, x)
        # the fold as the kernel does it (column sums x W) = the cell sums of hog.c:713-724 per patch
        colsum = rng.random(64).astype(np.float32)
        cells = colsum.astype(np.float64) @ W.astype(np.float64)
        for seg in range(3):
            for cx in range(C):
                direct = 0.0
                for x, a in enumerate(lanes):
                    if a["active"] and a["seg"] == seg:
                        b, w1, w2 = cell_weights(a["col"], cell, C)
                        if b == cx:
                            direct += float(colsum[x]) * float(w1)
                        if b + 1 == cx:
                            direct += float(colsum[x]) * float(w2)
                assert abs(direct - cells[seg * C + cx]) < 1e-9


def test_plan_packs_the_shipped_levels(built):
    # S = (55, 50, 40, 30) for the shipped parameters; 22 landmarks
    got = {cell: hog_plan(5, cell, 4, 22) for cell in (11, 10, 8, 6)}
    assert (got[11]["G"], got[11]["P"]) == (9, 8)           # nine 55-column patches in eight passes (round 6: groups of up to 12); 22 landmarks = 2 x 9 + 4 whole ones: 20 passes
    assert (got[11]["Gt"], got[11]["Pt"]) == (4, 4)         # ... the tail group's four patches one per pass: cuts that save no pass are not made
    assert (got[10]["G"], got[10]["P"]) == (5, 4)           # five 50-column patches in four passes
    assert (got[8]["G"], got[8]["P"]) == (3, 2)             # three 40-column patches in two passes
    assert (got[6]["G"], got[6]["P"]) == (2, 1)             # 30 columns: two patches per pass
    # 9 orientations ("31-bin" HOG, hog.c:212-215): packed since round 3 -- the lane plan does not depend on the orientation count
    # (18 bin rows are two matrix-core row tiles of the band fold), so it is the 4-orientation plan of the same cell size
    for cell in (11, 10, 8, 6):
        p9 = hog_plan(5, cell, 9, 22)
        assert p9 is not None and (p9["G"], p9["P"]) == (got[cell]["G"], got[cell]["P"])
        assert np.array_equal(p9["lane_tab"], got[cell]["lane_tab"]) and np.array_equal(p9["wb"], got[cell]["wb"])
    assert hog_plan(5, 11, 6, 22) is None                   # 6 orientations: no packed instance (its sector shortcut fails the exhaustive check)


@pytest.mark.parametrize("cell,L", [(11, 22), (10, 22), (8, 22), (6, 22), (10, 68), (8, 68), (6, 7), (10, 3)])
def test_cut_flags_match_the_lane_tables(cell, L):
    """Round 4 (CELLS form of the packed launch): a landmark's raw cell histograms arrive in two parts exactly when its patch
    appears in two passes of the plan -- the flags sdm_desc.hip reads must say so for every landmark, main groups and tail."""
    plan = hog_plan(5, cell, 4, L)
    assert plan is not None
    G, P, n_main, Gt, Pt = plan["G"], plan["P"], plan["n_main"], plan["Gt"], plan["Pt"]
    want = np.zeros(L, np.int32)

    def slots_in_two_passes(passes):
        seen = {}
        for pt in passes:
            for d in plan["lane_tab"][pt]:
                if (int(d) >> 17) & 1:
                    seen.setdefault(int(d) & 0xff, set()).add(pt)
        return [s for s, v in seen.items() if len(v) > 1], seen

    two, seen = slots_in_two_passes(range(P))
    assert all(len(v) <= 2 for v in seen.values())              # a patch (S <= 64 columns) is cut at most once
    for g in range(n_main):
        for s in two:
            want[g * G + s] = 1
    two_t, seen_t = slots_in_two_passes(range(P, P + Pt))
    assert all(len(v) <= 2 for v in seen_t.values())
    for s in two_t:
        want[n_main * G + s] = 1
    assert np.array_equal(plan["cut"], want)
    # the first pass of a cut patch is the one holding its column 0 (part 0), the second does not (part 1)
    for passes in (range(P), range(P, P + Pt)):
        for pt in passes:
            first_bits = (int(plan["pass_info"][pt][3]) >> 24) & 7
            for seg in range(3):
                slot = int(plan["pass_info"][pt][seg])
                if slot < 0:
                    continue
                cols = [(int(d) >> 8) & 0xff for d in plan["lane_tab"][pt] if ((int(d) >> 17) & 1) and (int(d) & 0xff) == slot and ((int(d) >> 20) & 3) == seg]
                assert ((first_bits >> seg) & 1) == (1 if 0 in cols else 0)
