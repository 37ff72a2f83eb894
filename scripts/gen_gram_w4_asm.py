#!/usr/bin/env python3
"""Writes superviseddescent_amd/csrc/sdm_gram_w4_asm.inc: the instruction stream of the four-wave float16-piece product
(sdm_gram_bf16.hip: syrk_tn_split_w4_kernel = A^T A / A^T b of regressors.hpp:208,225; syrk_update_f16_w4_kernel = the
Cholesky's trailing update), one wave per SIMD, every register and every wait placed by hand.

Why a generator and not C++: the compiler-scheduled form of this loop (kept in scripts/experiments/gram_w4_cxx.patch) needs 498 of
the wave's 512 registers for 352 live ones (sixteen unrolled steps fragment the tuple allocation), copies accumulator tiles around
the fold, spills, and drains the load queue at the loop header.  Here the register map is fixed:

    a[0:127]    acc    eight 32 x 32 accumulator tiles (tile t = 4 m + n: rows 32 m.., columns 32 n.. of the wave's 64 x 128)
    v[0:127]    tot    second accumulator level (Gram) / the tile of C read ahead (update)
    v[128:191]  fa     the wave's own rows: [stage 0..3][piece 0..1][row tile 0..1] x 4 registers, straight from the planes
    v[192:223]  fb     the column operand from LDS: [column tile 0..3][piece 0..1] x 4 registers (ONE set, see below)
    v[224:231]  temporaries;  v232 va  v233 vb  v234 baddr  v235 vc (lane offsets: rows, columns, LDS fragment, C)
    s[64:..]    pointers and counters (copied from the operands: they are advanced)

One step = one 16-row slab S = 24 matrix instructions (per accumulator tile: low x high, high x low, high x high -- the
eight-wave kernel's order).  The wave issues in order and has its SIMD to itself, so whatever is not a matrix instruction is
placed BEHIND one -- a load or an LDS read and a few scalar instructions at a time -- and issues while the pipe works on it:
    wait lgkmcnt(0): low column pieces of S
    products 0-7    acc[t] += fa[S][0][m] x fb[n][1]      behind 0-3: ds_read_b128 high column pieces of S (slot S & 3)
                                                          behind 4-5: global_load_dwordx4 rows of slab S+3, piece 0 -> stage (S-1)&3
                    in a fold step also the 16 read + add pairs of the folded tile, whose own product is the last of the eight,
                    from the matrix core's zero operand
    wait vmcnt(10): this wave's column pieces of S+1;  s_barrier: everybody's are there, slot (S-1)&3 is read out
    wait lgkmcnt(0): high column pieces of S
    products 8-15   acc[t] += fa[S][1][m] x fb[n][0]      behind 8-9: LDS-direct loads of slab S+3 -> slot (S-1)&3
                                                          behind 10-13: ds_read_b128 low column pieces of S+1
    products 16-23  acc[t] += fa[S][0][m] x fb[n][0]      behind 16-17: global_load_dwordx4 rows of slab S+4, piece 1 -> stage S&3
Vector-memory order per step: rows p0 (S+3) x 2, columns (S+3) x 2, rows p1 (S+4) x 2; "at most 10 outstanding" behind the
step's first two therefore means: the columns of S+1, and with them the rows of S+1, have landed.
Loads beyond the last slab read on into the planes' padding (sdm_gram_bf16x3_plane_bytes) and are never multiplied: the number
of loads per step, and with it every vmcnt, is static.

The second accumulator level is staggered: tile t is folded at the step behind slab 2 t + 1 of every 16 (256-row chunks as in
rounds 2-5; a tile's first chunk is shorter), one tile every other step.  The tile order of a step's three product groups is
rotated so that (a) the folded tile's restart is the last product of its group, (b) the tile folded NEXT is written last seven
matrix instructions before the step ends (its read-back must not follow its last product closely), (c) two products into one tile
are never closer than seven instructions (checked below).
"""
import os
import sys

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "superviseddescent_amd", "csrc", "sdm_gram_w4_asm.inc")

ACC, TOT, FA, FB, TMP = 0, 0, 128, 192, 224
VA, VB, VBADDR, VC = 232, 233, 234, 235
S_UA0, S_UA1, S_UB0, S_UB1, S_STEP, S_NSLABS, S_S, S_LDS, S_T = 64, 66, 68, 70, 72, 74, 75, 76, 77
S_CP, S_LDC1, S_LDC5, S_UNSCALE = 78, 80, 82, 84
SLOT_BYTES = 8192
N_TMP = 8


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
    def __init__(self):
        self.lines = []
        self.mfma_tiles = []          # tile of every matrix instruction in issue order (the loop body only), for the spacing check

    def op(self, text, comment=None):
        self.lines.append((text, comment))

    def note(self, comment):
        self.lines.append((None, comment))

    def mfma(self, t, a, b, zero=False, track=True):
        self.op(f"v_mfma_f32_32x32x16_f16 {acc(t)}, {a}, {b}, {'0' if zero else acc(t)}")
        if track:
            self.mfma_tiles.append(t)


def bump(s, lo):
    s.op(f"s_add_u32 s{lo}, s{lo}, s{S_STEP}")
    s.op(f"s_addc_u32 s{lo + 1}, s{lo + 1}, s{S_STEP + 1}")


def load_a(s, st, p, m):
    base = S_UA0 if p == 0 else S_UA1
    s.op(f"global_load_dwordx4 {fa(st, p, m)}, v{VA}, s[{base}:{base + 1}]" + (" offset:512" if m else ""))
    if m == 1:
        bump(s, base)


def set_m0(s, slot, p):
    s.op(f"s_add_u32 m0, s{S_LDS}, {slot * SLOT_BYTES + p * 4096}")


def load_b(s, p):
    base = S_UB0 if p == 0 else S_UB1
    s.op(f"global_load_lds_dwordx4 v{VB}, s[{base}:{base + 1}]")
    bump(s, base)


def read_b(s, p, slot, n):
    s.op(f"ds_read_b128 {fb(n, p)}, v{VBADDR} offset:{slot * SLOT_BYTES + p * 4096 + n * 512}")


def fold_at(q, mode):
    """tile folded at step q of 16 (behind the previous slab), None, or 'all'"""
    if mode == "none":
        return None
    if mode == "all":
        return "all" if q == 0 else None
    qp = (q + 15) % 16
    return (qp >> 1) if qp & 1 else None


def step(s, q, mode):
    """One slab.  A wave issues in order and alone on its SIMD: whatever is not a matrix instruction is placed BEHIND one, a
    load or an LDS read and a few scalar instructions at a time, so that it issues while the matrix pipe works on that one."""
    st = q & 3
    tf = fold_at(q, mode)
    # rotation of the tile order: the fold step 2 k + 2 (tile k) starts at k + 1 (tile k last), the step before it, 2 k + 1, at k
    # (tile k first: written last seven matrix instructions before the step ends) -- both are q >> 1
    r = (q >> 1) & 7 if mode == "stagger" else 0
    order = [(r + k) & 7 for k in range(8)]
    s.note(f"---- step {q}: stage {st}, tile order from {r}" + (f", folds tile {tf}" if tf is not None else ""))
    s.op("s_waitcnt lgkmcnt(0)", "low column pieces of this slab")
    if tf == "all":
        for t in range(8):
            for e in range(16):
                s.op(f"v_accvgpr_read_b32 v{TMP + e % N_TMP}, {acc(t, e)}")
                s.op(f"v_add_f32 {tot(t, e)}, {tot(t, e)}, v{TMP + e % N_TMP}")
    for k in range(8):
        t = order[k]
        fold_this = tf == "all" or (tf is not None and k == 7)
        if tf is not None and tf != "all" and k == 7:
            assert t == tf
        s.mfma(t, fa(st, 0, t >> 2), fb(t & 3, 1), zero=fold_this)
        if k < 4:
            read_b(s, 0, q & 3, k)                               # high pieces of THIS slab (their registers were last read by the previous slab)
        elif k < 6:
            load_a(s, (q + 3) & 3, 0, k - 4)                     # rows of slab S + 3, piece 0 -> the stage the previous slab has left
        elif k == 6:
            set_m0(s, (q + 3) & 3, 0)
        if tf is not None and tf != "all" and k < 7:
            es = list(range(16 * k // 7, 16 * (k + 1) // 7))
            for e in es:
                s.op(f"v_accvgpr_read_b32 v{TMP + e % N_TMP}, {acc(tf, e)}")
            for e in es:
                s.op(f"v_add_f32 {tot(tf, e)}, {tot(tf, e)}, v{TMP + e % N_TMP}")
    s.op("s_waitcnt vmcnt(10)", "this wave's column pieces of the next slab")
    s.op("s_barrier", "everybody's are there; everybody has read slab S - 1's slot for the last time")
    s.op("s_waitcnt lgkmcnt(0)", "high column pieces of this slab")
    for k in range(8):
        t = order[k]
        s.mfma(t, fa(st, 1, t >> 2), fb(t & 3, 0))
        if k == 0:
            load_b(s, 0)                                         # slab S + 3 -> slot (S - 1) & 3
            set_m0(s, (q + 3) & 3, 1)
        elif k == 1:
            load_b(s, 1)
        elif k < 6:
            read_b(s, 1, (q + 1) & 3, k - 2)                     # low pieces of the NEXT slab over the registers the first eight products have consumed
    for k in range(8):
        t = order[k]
        s.mfma(t, fa(st, 0, t >> 2), fb(t & 3, 0))
        if k < 2:
            load_a(s, st, 1, k)                                  # rows of slab S + 4, piece 1 -> this stage (its piece 1 is consumed)


def c_walk(s, body):
    """the wave's 64 x 128 of C, row by row: body(m, e) emits the four column tiles' work at the running row pointer"""
    for m in range(2):
        for e in range(16):
            body(m, e)
            if (m, e) != (1, 15):
                inc = S_LDC5 if (e & 3) == 3 else S_LDC1
                s.op(f"s_add_u32 s{S_CP}, s{S_CP}, s{inc}")
                s.op(f"s_addc_u32 s{S_CP + 1}, s{S_CP + 1}, s{inc + 1}")


def generate(mode, update):
    s = Stream()
    s.note(f"{'trailing update C -= P^T P' if update else 'Gram tile'}; second level: {mode}")
    # operands -> the fixed registers
    for dst, name in ((S_UA0, "ua0"), (S_UA1, "ua1"), (S_UB0, "ub0"), (S_UB1, "ub1"), (S_STEP, "step"), (S_CP, "cp"), (S_LDC1, "ldc1"), (S_LDC5, "ldc5")):
        s.op(f"s_mov_b64 s[{dst}:{dst + 1}], %[{name}]")
    for dst, name in ((S_NSLABS, "nslabs"), (S_LDS, "lds"), (S_UNSCALE, "unscale")):
        s.op(f"s_mov_b32 s{dst}, %[{name}]")
    for dst, name in ((VA, "va"), (VB, "vb"), (VBADDR, "baddr"), (VC, "vc")):
        s.op(f"v_mov_b32 v{dst}, %[{name}]")
    s.op(f"s_mov_b32 s{S_S}, 0")
    if update:
        # this wave's part of C is requested before anything else (it lands under the products; its registers are the second level's)
        s.op("s_cmp_eq_u32 %[writes], 0")
        s.op("s_cbranch_scc1 L_noc_%=")
        s.op(f"s_mov_b64 s[{S_T + 9}:{S_T + 10}], s[{S_CP}:{S_CP + 1}]", "(keep the row pointer for the store walk)")

        def cload(m, e):
            for n in range(4):
                s.op(f"global_load_dword {tot(4 * m + n, e)}, v{VC}, s[{S_CP}:{S_CP + 1}] offset:{128 * n}")
        c_walk(s, cload)
        s.op(f"s_mov_b64 s[{S_CP}:{S_CP + 1}], s[{S_T + 9}:{S_T + 10}]")
        s.op("L_noc_%=:")
    else:
        for i in range(128):
            s.op(f"v_mov_b32 v{TOT + i}, 0")
    for i in range(128):
        s.op(f"v_accvgpr_write_b32 a{i}, 0")
    # the issue order of the steady state from the start: rows (0), columns (0), rows (1), columns (1), rows (2), columns (2), rows (3)
    s.note("---- prologue")
    load_a(s, 0, 1, 0)
    load_a(s, 0, 1, 1)
    for k in range(3):
        load_a(s, k, 0, 0)
        load_a(s, k, 0, 1)
        set_m0(s, k, 0)
        s.op("s_nop 0")
        load_b(s, 0)
        set_m0(s, k, 1)
        s.op("s_nop 0")
        load_b(s, 1)
        load_a(s, k + 1, 1, 0)
        load_a(s, k + 1, 1, 1)
    s.op("s_waitcnt vmcnt(14)", "rows and column pieces of slab 0")
    s.op("s_barrier")
    for n in range(4):
        read_b(s, 1, 0, n)
    s.op("L_loop_%=:")
    n_before = len(s.mfma_tiles)
    for q in range(16):
        step(s, q, mode)
        if q & 3 == 3:
            s.op(f"s_add_u32 s{S_T}, s{S_S}, {q + 1}")
            s.op(f"s_cmp_ge_i32 s{S_T}, s{S_NSLABS}")
            if q < 15:
                s.op("s_cbranch_scc1 L_done_%=")
            else:
                s.op(f"s_mov_b32 s{S_S}, s{S_T}")
                s.op("s_cbranch_scc0 L_loop_%=")
    s.op("L_done_%=:")
    s.op("s_waitcnt vmcnt(0) lgkmcnt(0)", "every load has landed (the LDS-direct ones past the end too) before the LDS is given back")
    # spacing check over the loop body, cyclically
    seq = s.mfma_tiles[n_before:]
    last = {}
    for i, t in enumerate(seq + seq):
        if t in last:
            assert i - last[t] >= 7, (mode, i, t, i - last[t])
        last[t] = i
    # ---- epilogue
    s.note("---- epilogue")
    for _ in range(3):
        s.op("s_nop 7", "(the last products' accumulators are read back below)")
    s.op("s_cmp_eq_u32 %[writes], 0")
    s.op("s_cbranch_scc1 L_end_%=")

    def cstore(m, e):
        for n in range(4):
            t = 4 * m + n
            tmp = f"v{TMP + (4 * e + n) % N_TMP}"
            s.op(f"v_accvgpr_read_b32 {tmp}, {acc(t, e)}")
            if update:
                s.op(f"v_fma_f32 {tmp}, -{tmp}, s{S_UNSCALE}, {tot(t, e)}", "(the scale is a power of two: the product is exact)")
            else:
                s.op(f"v_add_f32 {tmp}, {tot(t, e)}, {tmp}")
                s.op(f"v_mul_f32 {tmp}, s{S_UNSCALE}, {tmp}")
            s.op(f"global_store_dword v{VC}, {tmp}, s[{S_CP}:{S_CP + 1}] offset:{128 * n}")
    c_walk(s, cstore)
    s.op("L_end_%=:")
    return s


def render(name, s):
    out = [f"#define {name} \\"]
    for text, comment in s.lines:
        if text is None:
            out.append(f"    /* {comment} */ \\")
        else:
            c = f"   /* {comment} */" if comment else ""
            out.append(f'    "{text}\\n"{c} \\')
    out.append('    ""')
    return "\n".join(out)


def main():
    clob = [f'"v{i}"' for i in range(236)] + [f'"a{i}"' for i in range(128)] + [f'"s{i}"' for i in range(64, 90)] + ['"scc"', '"vcc"', '"memory"']
    parts = ["// GENERATED by scripts/gen_gram_w4_asm.py -- edit the generator, not this file.  The instruction streams of the four-wave",
             "// float16-piece product kernels of sdm_gram_bf16.hip (register map, step layout and waits: see the generator's header).",
             "// clang-format off", ""]
    parts.append(render("SDM_GRAM_W4_ASM", generate("stagger", False)))
    parts.append("")
    parts.append(render("SDM_UPDATE_W4_ASM", generate("none", True)))
    parts.append("")
    parts.append("#define SDM_GRAM_W4_CLOBBERS " + ", ".join(clob))
    parts.append("")
    with open(OUT, "w") as f:
        f.write("\n".join(parts))
    print("wrote", os.path.normpath(OUT), sum(len(p.splitlines()) for p in parts), "lines")


if __name__ == "__main__":
    sys.exit(main())
