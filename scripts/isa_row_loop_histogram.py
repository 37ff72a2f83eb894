"""Instruction histogram of the row loop of hog_packed_kernel (two pixel rows per trip), from the compiler's own assembly:
    python scripts/isa_row_loop_histogram.py > profiles/r02_isa_row_loop_histogram.txt
Compiles superviseddescent_amd/csrc/sdm_hog_fast.hip to gfx950 assembly, takes the innermost row loop of the packed kernel
(the basic blocks between its loop header and back edge, band folds included) and counts mnemonics per class."""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "superviseddescent_amd", "csrc", "sdm_hog_fast.hip")
with tempfile.TemporaryDirectory() as d:
    out = os.path.join(d, "k.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-w", "-S",
                           "--cuda-device-only", "-o", out, src], stderr=subprocess.DEVNULL)
    lines = open(out).read().splitlines()
start = next(i for i, l in enumerate(lines) if re.match(r"^_ZN12_GLOBAL__N_117hog_packed_kernel.*:", l))
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
body = lines[start:end]
# the row loop: the "Inner Loop Header: Depth=2" block with the most instructions up to its back edge
best = None
for i, l in enumerate(body):
    if "Inner Loop Header: Depth=2" in l:
        # the label is on this line or on the line above the comment
        m = re.match(r"^(\.LBB\d+_\d+):", body[i]) or re.match(r"^(\.LBB\d+_\d+):", body[i - 1])
        if not m:
            continue
        lab = m.group(1)
        for j in range(len(body) - 1, i, -1):
            if re.search(r"s_c?branch\w*\s+%s\b" % re.escape(lab), body[j]):
                if best is None or j - i > best[1] - best[0]:
                    best = (i, j)
                break
i0, i1 = best
ins = [l.split()[0] for l in body[i0:i1 + 1] if l.startswith("\t") and not l.strip().startswith((";", "."))]
fold = [k for k in ins if k.startswith("v_mfma")]
cnt = collections.Counter(ins)


def cls(m):
    if m.startswith("v_mfma"): return "matrix (v_mfma)"
    if m.startswith("v_"): return "vector (v_*)"
    if m.startswith("s_waitcnt") or m.startswith("s_nop"): return "scalar: waits / nops"
    if m.startswith("s_load") : return "scalar memory (s_load)"
    if m.startswith("s_"): return "scalar ALU / branch"
    if m.startswith("ds_"): return "LDS (ds_*)"
    if m.startswith("buffer_") or m.startswith("global_"): return "vector memory"
    return "other"


per = collections.Counter()
for m, n in cnt.items():
    per[cls(m)] += n
print("hog_packed_kernel<4,5>: row loop, one trip = TWO pixel rows; the two inlined band folds (16 v_mfma each, taken once per")
print("cell row, i.e. every ~%s rows) are part of the listing -- static counts, lines %d..%d of the kernel\n" % ("cell", i0, i1))
for k, n in sorted(per.items(), key=lambda kv: -kv[1]):
    print("%-28s %4d" % (k, n))
print("\nper mnemonic:")
for m, n in sorted(cnt.items(), key=lambda kv: (-kv[1], kv[0])):
    print("  %-28s %4d" % (m, n))
