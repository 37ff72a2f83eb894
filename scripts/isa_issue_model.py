"""Compute-side roofline of the pixel kernel ("sum of class count x measured issue clocks / SIMD clocks"):
compiles csrc/sdm_hog_fast.hip to gfx950 assembly, takes the four raw-cells instances the shipped RCR-22 model runs (cell sizes
11 / 10 / 8 / 6; their pixel-row loops are fully unrolled, so a static count IS the dynamic count of a pass), classifies every
instruction between the first and the last image load + the band folds behind them, and prices the classes with the issue clocks
measured on this chip (profiles/r03_ubench_valu_rates.txt, 8 waves per SIMD column; v_mfma_f32_16x16x4_f32: 8 passes = 32 clocks).
Writes profiles/r04_issue_model.json; bench.py puts the resulting fraction beside the HBM one.
usage: python scripts/isa_issue_model.py   (needs hipcc; no GPU)"""
import json, os, re, subprocess, sys
from collections import Counter
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ASM = "/tmp/isa/issue_model.s"
os.makedirs("/tmp/isa", exist_ok=True)
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-w", "-S",
                       "--cuda-device-only", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "superviseddescent_amd/csrc/sdm_hog_fast.hip"), "-o", ASM])
FAST = {"v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32",
        "v_lshrrev_b32", "v_lshlrev_b32", "v_ashrrev_i32", "v_mov_b32", "v_max_f32", "v_min_f32", "v_max_i32", "v_min_i32", "v_max_u32", "v_min_u32", "v_accvgpr_write_b32"}
TRANS = {"v_sqrt_f32", "v_rsq_f32", "v_rcp_f32", "v_rcp_iflag_f32", "v_exp_f32", "v_log_f32"}
TRANS64 = {"v_sqrt_f64", "v_rsq_f64", "v_rcp_f64"}
RATE = {"fast": 2.8, "full": 4.4, "trans": 8.2, "trans64": 16.4}      # clocks per wave-instruction per SIMD (r03_ubench_valu_rates.txt, w8)
def cls(op):
    base = re.sub(r"_(e32|e64|dpp|sdwa)$", "", op)
    if base in FAST and not op.endswith("_dpp"): return "fast"
    if base in TRANS: return "trans"
    if base in TRANS64: return "trans64"
    return "full"
src = open(ASM).read()
# geometry of the shipped levels: (cell, S, passes per face with the one-pass-per-wave plan)
LEVELS = {11: (55, 22), 10: (50, 18), 8: (40, 15), 6: (30, 11)}
out = {"rates_clocks_per_wave_instruction": RATE, "mfma_clocks": {"f32_16x16x4": 32, "f16_16x16x32": 17}, "levels": {}}
for m in re.finditer(r"^(_Z\S*hog_packed_kernelILi4ELi5ELi(\d+)ELb1ELb1E\S*):", src, re.M):
    cell = int(m.group(2))
    if cell not in LEVELS: continue
    body = src[m.end():src.index("s_endpgm", m.end())]
    ins = [l.strip().split()[0] for l in body.split("\n") if l.strip() and not l.strip().startswith((".", ";", "//")) and not l.strip().endswith(":")]
    first = next(i for i, o in enumerate(ins) if o.startswith("buffer_load_ushort"))
    loop = ins[first:]                      # the pass from its first image load to the end of the kernel: rows + folds + cell stores
    setup = ins[:first]
    def price(seq):
        c = Counter()
        for o in seq:
            if o.startswith("v_mfma"): c["mfma"] += 1
            elif o.startswith("v_"): c[cls(o)] += 1
            elif o.startswith("s_"): c["salu"] += 1
            elif o.startswith("ds_"): c["lds"] += 1
            elif o.startswith(("buffer_", "global_")): c["vmem"] += 1
        valu = sum(c[k] * RATE[k] for k in RATE)
        return dict(c), valu
    lc, lvalu = price(loop)
    sc, svalu = price(setup)
    S, ppf = LEVELS[cell]
    # the pair fold site and the two single-fold sites it replaces are both in the code (5 sites); a pass executes 3 single folds + 1 pair
    # fold = 4/5 of the static matrix instructions.  f32 16x16x4: 32 clocks each, the vector pipe of the SIMD runs at 40 % meanwhile;
    # f16 16x16x32 (round 4: the folds on float16 pieces): 17 clocks, vector pipe at 58 % (profiles/r04_ubench_mfma_valu_overlap.txt)
    f16 = sum(1 for o in loop if o.startswith("v_mfma_f32_16x16x32_f16"))
    f32 = sum(1 for o in loop if o.startswith("v_mfma_f32_16x16x4_f32") or o.startswith("v_mfma_f32_16x16x4f32"))
    mfma_clocks = (f16 * 17 + f32 * 32) * 4 // 5
    valu_loss = (f16 * 17 * 0.42 + f32 * 32 * 0.60) * 4 / 5
    out["levels"][str(cell)] = {"rows": S, "passes_per_face": ppf, "row_loop_and_folds": lc, "setup_static": sc,
                                "valu_clocks_row_loop_and_folds": lvalu, "valu_clocks_per_row": lvalu / S,
                                "valu_clocks_setup_static_upper_bound": svalu, "mfma_clocks_per_pass": mfma_clocks,
                                "valu_clocks_lost_beside_mfma_per_pass": valu_loss, "mfma_static": {"f16_16x16x32": f16, "f32_16x16x4": f32}}
json.dump(out, open(os.path.join(ROOT, "profiles", "r04_issue_model.json"), "w"), indent=1)
for k, v in out["levels"].items():
    print(k, {kk: vv for kk, vv in v.items() if not isinstance(vv, dict)}, v["row_loop_and_folds"])
