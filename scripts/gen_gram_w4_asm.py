#!/usr/bin/env python3
"""Writes superviseddescent_amd/csrc/sdm_gram_w4_asm.inc: the instruction streams of the four-wave float16-piece product kernels
(sdm_gram_bf16.hip: syrk_tn_split_w4_kernel = A^T A / A^T b of regressors.hpp:208,225, one wave per SIMD; syrk_update_f16_w4_kernel =
the Cholesky's trailing update, two waves per SIMD), every register and every wait placed by hand.

Why a generator and not C++: the compiler-scheduled form of this loop (kept in scripts/experiments/gram_w4_cxx.patch) needs 498 of
the wave's 512 registers for 352 live ones (sixteen unrolled steps fragment the tuple allocation), copies accumulator tiles around
the fold, spills, and drains the load queue at the loop header.  Here the register maps are fixed (MAPS below).  The Gram kernel's:

    a[0:127]    acc    eight 32 x 32 accumulator tiles (tile t = 4 m + n: rows 32 m.., columns 32 n.. of the wave's 64 x 128)
    v[0:127]    tot    second accumulator level
    v[128:191]  fa     the wave's own rows: [stage 0..3][piece 0..1][row tile 0..1] x 4 registers, straight from the planes
    v[192:223]  fb     the column operand from LDS: [column tile 0..3][piece 0..1] x 4 registers (ONE set, see below)
    v[224:231]  temporaries;  v232 va  v233 vb  v234 baddr  v235 vc (lane offsets: rows, columns, LDS fragment, C)
    s[64:..]    pointers and counters (copied from the operands: they are advanced)

The trailing update's (one accumulator level at K <= 512, C read behind the loop): fa in v[0:63], fb in v[64:95], temporaries and lane
offsets in v[96:107], sixteen landing registers for C in v[108:123] -- 124 + 128 registers, two waves per SIMD.

One step = one 16-row slab S = 24 matrix instructions (per accumulator tile: low x high, high x low, high x high -- the
eight-wave kernel's order).  The wave issues in order and has its SIMD to itself, so whatever is not a matrix instruction is
placed BEHIND one -- a load or an LDS read and a few scalar instructions at a time -- and issues while the pipe works on it:
    wait: low column pieces of S
    products 0-7    acc[t] += fa[S][0][m] x fb[n][1]      behind them: ds_read_b128 high column pieces of S (slot S & 3),
                                                          global_load_dwordx4 rows of slab S+3, piece 0 -> stage (S-1)&3
                    in a fold step also the 16 read + add pairs of the folded tile, whose own product is the last of the eight,
                    from the matrix core's zero operand
    wait: this wave's column pieces of S+1;  s_barrier: everybody's are there, slot (S-1)&3 is read out
    wait: high column pieces of S
    products 8-15   acc[t] += fa[S][1][m] x fb[n][0]      behind them: LDS-direct loads of slab S+3 -> slot (S-1)&3,
                                                          ds_read_b128 low column pieces of S+1
    products 16-23  acc[t] += fa[S][0][m] x fb[n][0]      behind them: global_load_dwordx4 rows of slab S+4, piece 1 -> stage S&3
Loads beyond the last slab read on into the planes' padding (sdm_gram_bf16x3_plane_bytes) and are never multiplied: the number
of loads per step is static.  The waits are COMPUTED: the generator keeps the order in which vector-memory loads and LDS reads
were issued (both return in order) and writes "at most n outstanding" with n = what was issued behind the operation waited for
(class Stream); the loop body is emitted for two consecutive trips and the second's text (the steady state) must equal the
first's.

The second accumulator level is staggered: tile t is folded at the step behind slab 2 t + 1 of every 16 (256-row chunks as in
rounds 2-5; a tile's first chunk is shorter), one tile every other step.  The tile order of a step's three product groups is
rotated so that (a) the folded tile's restart is the last product of its group, (b) the tile folded NEXT is written last seven
matrix instructions before the step ends (its read-back must not follow its last product closely), (c) two products into one tile
keep their distance (checked).

The generator can emit narrower variants of the stream in the same code object, chosen per wave by a `variant` operand (VARIANTS
below): two / one column tiles for a right-hand-side tile column with <= 64 / <= 32 columns (the other products are zeros), and an
idle stream for a wave whose 64 rows lie below the diagonal (it loads its share of the column pieces, keeps the barriers and
multiplies nothing).  Measured on one box, Gram stage at 100 000 rows: RCR-22 19.1-19.4 ms with the full stream for every wave,
19.8-19.9 ms with the variants; RCR-68 177.4 against 180.9 ms; factor + solve unchanged (31.8 / 10.4 / 3.5 ms either way) -- the few
workgroups that get shorter do not shorten the launch (its length is whole rounds of workgroups), and their steps without
products are bound by the loads.  Shipped: the full stream only.
"""
import os
import sys

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "superviseddescent_amd", "csrc", "sdm_gram_w4_asm.inc")

# register maps (vector registers; the accumulators are a[0:127] in both).  "wide": the map of the header, 236 + 128 registers = one
# wave per SIMD.  "slim" (the trailing update, round 6): no second level and no tile of C held through the loop -- the rows' fragments,
# the column fragments, temporaries and lane offsets in v[0:107], sixteen landing registers for C in v[108:123]: 124 + 128 registers,
# TWO waves per SIMD, i.e. two workgroups per compute unit -- one's prologue (first operands) and epilogue (read C, subtract, store:
# nothing a lone wave can overlap with products) run under the other's products.
MAPS = {"wide": dict(TOT=0, FA=128, FB=192, TMP=224, VA=232, VB=233, VBADDR=234, VC=235, CT=None, NV=236),
        "slim": dict(TOT=None, FA=0, FB=64, TMP=96, VA=104, VB=105, VBADDR=106, VC=107, CT=108, NV=124)}
ACC = 0
TOT = FA = FB = TMP = VA = VB = VBADDR = VC = CT = NV = None


def use_map(name):
    globals().update(MAPS[name])


use_map("wide")
S_UA0, S_UA1, S_UB0, S_UB1, S_STEP, S_NSLABS, S_S, S_LDS, S_T = 64, 66, 68, 70, 72, 74, 75, 76, 77
S_CP, S_LDC1, S_LDC5, S_UNSCALE, S_CP0 = 78, 80, 82, 84, 86
SLOT_BYTES = 8192
N_TMP = 8
# (name, value of the `variant` operand, column tiles, idle); the first one is the fall-through.  All four:
#   (("n4", 0, 4, False), ("n2", 1, 2, False), ("n1", 2, 1, False), ("idle", 3, 4, True))
VARIANTS = (("n4", 0, 4, False),)


def acc(t, e=None):
    return f"a[{16 * t}:{16 * t + 15}]" if e is None else f"a{16 * t + e}"


def tot(t, e):
    return f"v{TOT + 16 * t + e}"


def fa(st, p, m):
    b = FA + 16 * st + 8 * p + 4 * m
    return f"v[{b}:{b + 3}]"


def fb(n, p):
    b = FB + 8 * n + 4 * p
    return f"v[{b}:{b + 3}]"


class Stream:
    """Instruction text + the two in-order queues the waits are counted on."""

    def __init__(self, suffix):
        self.lines = []
        self.suffix = suffix          # label suffix of this variant
        self.vm = []                  # tags of the vector-memory loads in issue order
        self.vm_done = 0              # loads [0, vm_done) are known to have landed
        self.lds = []                 # tags of the LDS reads in issue order
        self.lds_done = 0
        self.mfma_tiles = []

    def op(self, text, comment=None):
        self.lines.append((text, comment))

    def note(self, comment):
        self.lines.append((None, comment))

    def label(self, name):
        self.op(f"L_{name}_{self.suffix}_%=:")

    def ref(self, name):
        return f"L_{name}_{self.suffix}_%="

    def vmem(self, text, tag):
        self.op(text)
        self.vm.append(tag)

    def dsread(self, text, tag):
        self.op(text)
        self.lds.append(tag)

    def _wait(self, queue, done, tag, name, comment):
        idx = max(i for i, t in enumerate(queue) if t == tag)      # the last operation of that tag
        if idx < done:
            return done
        self.op(f"s_waitcnt {name}({len(queue) - 1 - idx})", comment)
        return idx + 1

    def wait_vm(self, tag, comment=None):
        self.vm_done = self._wait(self.vm, self.vm_done, tag, "vmcnt", comment)

    def wait_lds(self, tag, comment=None):
        self.lds_done = self._wait(self.lds, self.lds_done, tag, "lgkmcnt", comment)

    def mfma(self, t, a, b, zero=False):
        self.op(f"v_mfma_f32_32x32x16_f16 {acc(t)}, {a}, {b}, {'0' if zero else acc(t)}")
        self.mfma_tiles.append(t)


def bump(s, lo):
    s.op(f"s_add_u32 s{lo}, s{lo}, s{S_STEP}")
    s.op(f"s_addc_u32 s{lo + 1}, s{lo + 1}, s{S_STEP + 1}")


def load_a(s, slab, p, m):
    base = S_UA0 if p == 0 else S_UA1
    s.vmem(f"global_load_dwordx4 {fa(slab & 3, p, m)}, v{VA}, s[{base}:{base + 1}]" + (" offset:512" if m else ""), ("A", p, slab))
    if m == 1:
        bump(s, base)


def set_m0(s, slot, p):
    s.op(f"s_add_u32 m0, s{S_LDS}, {slot * SLOT_BYTES + p * 4096}")


def load_b(s, slab, p):
    base = S_UB0 if p == 0 else S_UB1
    s.vmem(f"global_load_lds_dwordx4 v{VB}, s[{base}:{base + 1}]", ("B", slab))
    bump(s, base)


def read_b(s, slab, p, n):
    s.dsread(f"ds_read_b128 {fb(n, p)}, v{VBADDR} offset:{(slab & 3) * SLOT_BYTES + p * 4096 + n * 512}", ("b", p, slab))


def fold_at(q, fold):
    """tile folded at step q of 16 (behind the previous slab), or None"""
    if not fold:
        return None
    qp = (q + 15) % 16
    return (qp >> 1) if qp & 1 else None


def place(n_products, emit_product, fillers):
    """n_products matrix instructions, the fillers (lists of emitters) spread behind them, one filler per product; more fillers than
    products: the rest behind the last one; no products (an idle wave): the fillers alone"""
    if n_products == 0:
        for f in fillers:
            for e in f:
                e()
        return
    for k in range(n_products):
        emit_product(k)
        for f in (fillers[k:k + 1] if k < n_products - 1 else fillers[k:]):
            for e in f:
                e()


def step(s, S, cfg):
    """One slab (S: its number counted from the start of the emitted stream; only S mod 16 enters the text)."""
    q = S & 15
    st = S & 3
    nt, idle, fold = cfg["ntiles"], cfg["idle"], cfg["fold"]
    live = [] if idle else [t for t in range(8) if (t & 3) < nt]
    tf = fold_at(q, fold)
    if tf is not None and tf not in live:
        tf = None
    # rotation of the tile order: the fold step 2 k + 2 (tile k) starts at k + 1 (tile k last), the step before it, 2 k + 1, at k
    # (tile k first: written last seven matrix instructions before the step ends) -- both are q >> 1
    r = (q >> 1) & 7 if fold else 0
    order = [t for t in [(r + k) & 7 for k in range(8)] if t in live]
    ns = list(range(nt)) if not idle else []
    s.note(f"---- step {q}: stage {st}" + (f", tile order from {r}" if live else "") + (f", folds tile {tf}" if tf is not None else ""))
    if not idle:
        s.wait_lds(("b", 1, S), "low column pieces of this slab")
        s.wait_vm(("A", 0, S), "this slab's rows, piece 0")
    # ---- first product group
    f1 = [[lambda n=n: read_b(s, S, 0, n)] for n in ns]          # high pieces of THIS slab (their registers were last read by the previous slab)
    if not idle:
        f1 += [[lambda: load_a(s, S + 3, 0, 0)], [lambda: load_a(s, S + 3, 0, 1)]]      # rows of slab S + 3, piece 0 -> the stage the previous slab has left
    f1 += [[lambda: set_m0(s, (S + 3) & 3, 0)]]

    def g1(k):
        t = order[k]
        last = tf is not None and k == len(order) - 1
        if last:
            assert t == tf
        s.mfma(t, fa(st, 0, t >> 2), fb(t & 3, 1), zero=last)
        if tf is not None and k < len(order) - 1:
            n_gaps = len(order) - 1
            es = list(range(16 * k // n_gaps, 16 * (k + 1) // n_gaps))
            if k == 0 and len(order) < 8:
                # with fewer than eight live tiles the folded tile's last product is fewer than seven matrix instructions back (one, with
                # two live tiles): the read-back of an accumulator wants ~19 wait states behind the instruction that wrote it
                s.op("s_nop 7")
                s.op("s_nop 7")
                s.op("s_nop 7")
            for c0 in range(0, len(es), N_TMP):      # (at most as many reads in flight as there are temporaries)
                chunk = es[c0:c0 + N_TMP]
                for e in chunk:
                    s.op(f"v_accvgpr_read_b32 v{TMP + e % N_TMP}, {acc(tf, e)}")
                for e in chunk:
                    s.op(f"v_add_f32 {tot(tf, e)}, {tot(tf, e)}, v{TMP + e % N_TMP}")
    place(len(order), g1, f1)
    s.wait_vm(("B", S + 1), "this wave's column pieces of the next slab")
    s.op("s_barrier", "everybody's are there; everybody has read slab S - 1's slot for the last time")
    if not idle:
        s.wait_lds(("b", 0, S), "high column pieces of this slab")
        s.wait_vm(("A", 1, S), "this slab's rows, piece 1")
    # ---- second product group
    f2 = [[lambda: load_b(s, S + 3, 0), lambda: set_m0(s, (S + 3) & 3, 1)], [lambda: load_b(s, S + 3, 1)]]      # slab S + 3 -> slot (S - 1) & 3
    f2 += [[lambda n=n: read_b(s, S + 1, 1, n)] for n in ns]                                                      # low pieces of the NEXT slab
    place(len(order), lambda k: s.mfma(order[k], fa(st, 1, order[k] >> 2), fb(order[k] & 3, 0)), f2)
    # ---- third product group
    f3 = []
    if not idle:
        f3 += [[lambda: load_a(s, S + 4, 1, 0)], [lambda: load_a(s, S + 4, 1, 1)]]      # rows of slab S + 4, piece 1 -> this stage (its piece 1 is consumed)
    place(len(order), lambda k: s.mfma(order[k], fa(st, 0, order[k] >> 2), fb(order[k] & 3, 0)), f3)


def c_rows():
    """the wave's 64 x 128 of C row by row: (m, e, increment of the row pointer behind it: 1 row, 5 rows, or None at the end)"""
    return [(m, e, None if (m, e) == (1, 15) else (S_LDC5 if (e & 3) == 3 else S_LDC1)) for m in range(2) for e in range(16)]


def advance_cp(s, inc):
    if inc is not None:
        s.op(f"s_add_u32 s{S_CP}, s{S_CP}, s{inc}")
        s.op(f"s_addc_u32 s{S_CP + 1}, s{S_CP + 1}, s{inc + 1}")


def emit_variant(s, cfg):
    """prologue + loop of one variant; leaves through the common epilogue"""
    idle, nt = cfg["idle"], cfg["ntiles"]
    # the issue order of the steady state from the start: rows p1 (0), then per slab k = 0..2: rows p0 (k), columns (k), rows p1 (k + 1)
    s.note("---- prologue")
    if not idle:
        load_a(s, 0, 1, 0)
        load_a(s, 0, 1, 1)
    for k in range(3):
        if not idle:
            load_a(s, k, 0, 0)
            load_a(s, k, 0, 1)
        set_m0(s, k, 0)
        s.op("s_nop 0")
        load_b(s, k, 0)
        set_m0(s, k, 1)
        s.op("s_nop 0")
        load_b(s, k, 1)
        if not idle:
            load_a(s, k + 1, 1, 0)
            load_a(s, k + 1, 1, 1)
    s.wait_vm(("B", 0), "column pieces of slab 0")
    s.op("s_barrier")
    if not idle:
        for n in range(nt):
            read_b(s, 0, 1, n)
    n_before = len(s.mfma_tiles)

    def trip(base):
        for q in range(16):
            step(s, base + q, cfg)
            if q & 3 == 3:
                s.op(f"s_add_u32 s{S_T}, s{S_S}, {q + 1}")
                s.op(f"s_cmp_ge_i32 s{S_T}, s{S_NSLABS}")
                if q < 15:
                    s.op(f"s_cbranch_scc1 {s.ref('done')}")
                else:
                    s.op(f"s_mov_b32 s{S_S}, s{S_T}")
                    s.op(f"s_cbranch_scc0 {s.ref('loop')}")
    s.label("loop")
    m1 = len(s.lines)
    trip(0)
    m2 = len(s.lines)
    trip(16)                               # the steady state: must read the same
    assert [t for t, _ in s.lines[m1:m2]] == [t for t, _ in s.lines[m2:]], (cfg["ntiles"], cfg["idle"], "the loop's waits are not stationary")
    del s.lines[m2:]
    s.label("done")
    # spacing of two products into one accumulator tile, cyclically over the loop body
    per_trip = 3 * len([t for t in range(8) if (t & 3) < nt]) * 16 if not idle else 0
    seq = s.mfma_tiles[n_before:][:per_trip]
    last = {}
    for i, t in enumerate(seq + seq):
        if t in last:
            assert i - last[t] >= min(7, 2 * nt - 1), (nt, i, t, i - last[t])
        last[t] = i
    s.op("s_branch L_epilogue_%=")


def c_dst_slim(i):
    """landing register of load i (of 128) of the wave's part of C in the slim map: the first 112 in one round (the fragment registers
    are dead behind the loop), the last 16 in a second"""
    if i < 96:
        return f"v{i}"
    if i < 112:
        return f"v{CT + i - 96}"
    return f"v{i - 112}"


def generate_update():
    """the trailing update on the slim register map (two waves per SIMD): the Gram loop without the second level, C read behind the loop"""
    use_map("slim")
    parts = []
    s0 = Stream("x")
    s0.note("trailing update C -= P^T P, 124 + 128 registers")
    entry(s0, False)
    parts.append(s0)
    s = Stream("n4")
    emit_variant(s, {"ntiles": 4, "idle": False, "fold": False})
    parts.append(s)
    e = Stream("e")
    epilogue_head(e, "every load has landed (the ones past the last slab too: their registers are reused below)")
    rows = c_rows()

    def loads(i0, i1):
        for i in range(i0, i1):
            m, ee, inc = rows[i // 4]
            n = i % 4
            e.op(f"global_load_dword {c_dst_slim(i)}, v{VC}, s[{S_CP}:{S_CP + 1}] offset:{128 * n}")
            if n == 3:
                advance_cp(e, inc)

    def finish(i0, i1):
        for i in range(i0, i1):
            m, ee, inc = rows[i // 4]
            n = i % 4
            tmp = f"v{TMP + i % N_TMP}"
            e.op(f"v_accvgpr_read_b32 {tmp}, {acc(4 * m + n, ee)}")
            e.op(f"v_fma_f32 {tmp}, -{tmp}, s{S_UNSCALE}, {c_dst_slim(i)}", "(the scale is a power of two: the product is exact)" if i == 0 else None)
            e.op(f"global_store_dword v{VC}, {tmp}, s[{S_CP0}:{S_CP0 + 1}] offset:{128 * n}")
            if n == 3 and inc is not None:
                e.op(f"s_add_u32 s{S_CP0}, s{S_CP0}, s{inc}")
                e.op(f"s_addc_u32 s{S_CP0 + 1}, s{S_CP0 + 1}, s{inc + 1}")
    e.note("---- C: rows 0-27 of this wave's 32 (x 4 per lane half) into the dead fragment registers + the landing registers")
    loads(0, 112)
    e.op("s_waitcnt vmcnt(0)")
    finish(0, 16)
    e.note("---- the last four rows into the registers just consumed; they land while the other 96 values are finished")
    loads(112, 128)
    finish(16, 112)
    e.op("s_waitcnt vmcnt(0)")
    finish(112, 128)
    e.op("L_end_%=:")
    parts.append(e)
    use_map("wide")
    return parts


def entry(s0, zero_tot):
    """operands -> the fixed registers, accumulators cleared"""
    for dst, name in ((S_UA0, "ua0"), (S_UA1, "ua1"), (S_UB0, "ub0"), (S_UB1, "ub1"), (S_STEP, "step"), (S_CP, "cp"), (S_LDC1, "ldc1"), (S_LDC5, "ldc5")):
        s0.op(f"s_mov_b64 s[{dst}:{dst + 1}], %[{name}]")
    for dst, name in ((S_NSLABS, "nslabs"), (S_LDS, "lds"), (S_UNSCALE, "unscale")):
        s0.op(f"s_mov_b32 s{dst}, %[{name}]")
    for dst, name in ((VA, "va"), (VB, "vb"), (VBADDR, "baddr"), (VC, "vc")):
        s0.op(f"v_mov_b32 v{dst}, %[{name}]")
    s0.op(f"s_mov_b32 s{S_S}, 0")
    s0.op(f"s_mov_b64 s[{S_CP0}:{S_CP0 + 1}], s[{S_CP}:{S_CP + 1}]", "(the row pointer of the store walk)")
    if zero_tot:
        for i in range(128):
            s0.op(f"v_mov_b32 v{TOT + i}, 0")
    for i in range(128):
        s0.op(f"v_accvgpr_write_b32 a{i}, 0")


def epilogue_head(e, why):
    e.op("L_epilogue_%=:")
    e.op("s_waitcnt vmcnt(0) lgkmcnt(0)", why)
    for _ in range(3):
        e.op("s_nop 7", "(the last products' accumulators are read back below)")
    e.op("s_cmp_ge_i32 %[row0], %[rowend]", "this wave's 64 rows are written iff they lie above the tile column's diagonal and inside the rows that exist")
    e.op("s_cbranch_scc1 L_end_%=")


def generate_gram():
    """the Gram tile on the wide register map: two accumulator levels, (tot + acc) * unscale stored"""
    use_map("wide")
    parts = []
    s0 = Stream("x")
    s0.note("Gram tile")
    entry(s0, True)
    for name, v, _ntiles, _idle in VARIANTS[1:]:
        s0.op(f"s_cmp_eq_u32 %[variant], {v}")
        s0.op(f"s_cbranch_scc1 L_entry_{name}_%=")
    parts.append(s0)
    for name, _v, ntiles, idle in VARIANTS:
        s = Stream(name)
        s.op(f"L_entry_{name}_%=:")
        s.note(f"==== variant {name}")
        emit_variant(s, {"ntiles": ntiles, "idle": idle, "fold": not idle})
        parts.append(s)
    e = Stream("e")
    epilogue_head(e, "every load has landed (the LDS-direct ones past the end too) before the LDS is given back")
    for m, ee, inc in c_rows():
        for n in range(4):
            t = 4 * m + n
            tmp = f"v{TMP + (4 * ee + n) % N_TMP}"
            e.op(f"v_accvgpr_read_b32 {tmp}, {acc(t, ee)}")
            e.op(f"v_add_f32 {tmp}, {tot(t, ee)}, {tmp}")
            e.op(f"v_mul_f32 {tmp}, s{S_UNSCALE}, {tmp}")
            e.op(f"global_store_dword v{VC}, {tmp}, s[{S_CP}:{S_CP + 1}] offset:{128 * n}")
        advance_cp(e, inc)
    e.op("L_end_%=:")
    parts.append(e)
    return parts


def render(name, parts):
    out = [f"#define {name} \\"]
    for s in parts:
        for text, comment in s.lines:
            if text is None:
                out.append(f"    /* {comment} */ \\")
            else:
                c = f"   /* {comment} */" if comment else ""
                out.append(f'    "{text}\\n"{c} \\')
    out.append('    ""')
    return "\n".join(out)


def main():
    def clobbers(nv):
        return [f'"v{i}"' for i in range(nv)] + [f'"a{i}"' for i in range(128)] + [f'"s{i}"' for i in range(64, 90)] + ['"scc"', '"vcc"', '"memory"']
    clob = clobbers(MAPS["wide"]["NV"])
    parts = ["// GENERATED by scripts/gen_gram_w4_asm.py -- edit the generator, not this file.  The instruction streams of the four-wave",
             "// float16-piece product kernels of sdm_gram_bf16.hip (register map, step layout, variants and how the waits are counted: see",
             "// the generator's header).",
             "// clang-format off", ""]
    parts.append(render("SDM_GRAM_W4_ASM", generate_gram()))
    parts.append("")
    parts.append(render("SDM_UPDATE_W4_ASM", generate_update()))
    parts.append("")
    parts.append("#define SDM_GRAM_W4_CLOBBERS " + ", ".join(clob))
    parts.append("#define SDM_UPDATE_W4_CLOBBERS " + ", ".join(clobbers(MAPS["slim"]["NV"])))
    parts.append("")
    out = sys.argv[1] if len(sys.argv) > 1 else OUT      # (tests/test_gram_stream_generator.py writes to a scratch path and compares)
    with open(out, "w") as f:
        f.write("\n".join(parts))
    print("wrote", os.path.normpath(out), sum(len(p.splitlines()) for p in parts), "lines")


if __name__ == "__main__":
    sys.exit(main())
