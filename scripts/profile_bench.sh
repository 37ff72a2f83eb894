#!/bin/bash
# Round profile: rocprofv3 --kernel-trace --stats of the default bench + separate PMC passes (HBM bytes, MFMA busy),
# summaries written to gpurun_out/profile_<tag>/ (copy the *.txt / *.csv you want judged into profiles/).
set -u
TAG=${1:-run}
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/gpurun_out/profile_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $REPO/bench.py --steps 20 --warmup 2 --no-cpu --train-rows 20000 --rcr68-shard 0"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- $CMD > $OUT/bench_under_trace.json 2> $OUT/trace_stderr.log
cp $OUT/trace/t_kernel_stats.csv $OUT/kernel_stats.csv 2>/dev/null
i=0
for pmc in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAVE_CYCLES"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $pmc --output-format csv -d $OUT/p$i -o pmc -- $CMD > /dev/null 2> $OUT/p${i}_stderr.log
done
# the full default command (RCR-22 headline + RCR-68 train 100k rows + RCR-68 detect shard): kernel stats only
CMD68="python $REPO/bench.py --steps 20 --warmup 2 --no-cpu"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace68 -o t -- $CMD68 > $OUT/bench_full_under_trace.json 2> $OUT/trace68_stderr.log
python - <<PY
import csv, glob, collections, re
out="$OUT"
def short(n):
    m=re.search(r'(\w+_kernel)', n); return m.group(1) if m else n.split("(")[0][:40]
try:
    with open(out+"/summary_full_bench.txt","w") as fh:
        fh.write("== rocprofv3 --kernel-trace --stats : python bench.py --steps 20 --warmup 2 --no-cpu  (RCR-22 train 100k + detect 4096, RCR-68 train 100k + detect 8192) ==\n")
        fh.write("%-32s %7s %13s %11s %7s\n" % ("kernel","calls","total_us","avg_us","%"))
        for r in list(csv.DictReader(open(out+"/trace68/t_kernel_stats.csv")))[:18]:
            fh.write("%-32s %7s %13.1f %11.2f %7.2f\n" % (short(r["Name"]), r["Calls"], float(r["TotalDurationNs"])/1e3, float(r["AverageNs"])/1e3, float(r["Percentage"])))
        geo=collections.defaultdict(list)
        for f in glob.glob(out+"/trace68/*kernel_trace.csv"):
            for r in csv.DictReader(open(f)):
                k=short(r["Kernel_Name"])
                if k in ("hog_packed_kernel","desc_kernel","apply_partial_kernel","apply_tiled_kernel","apply_tiled_f16_kernel","apply_reduce_kernel","syrk_tn_glds_kernel","syrk_tn_gldsw_kernel","syrk_tn_split_w4_kernel","syrk_update_f16_w4_kernel","syrk_tn_split_w8p_kernel","syrk_tn_bf16x3_w_kernel","split_planes_f16_kernel","backsolve_persistent_kernel"):
                    geo[(k,"x".join(r.get(c,"?") for c in ("Grid_Size_X","Grid_Size_Y","Grid_Size_Z")))].append((float(r["End_Timestamp"])-float(r["Start_Timestamp"]))/1e3)
        fh.write("\n== by launch geometry (avg over dispatches) ==\n%-32s %16s %7s %11s\n" % ("kernel","grid","calls","avg_us"))
        for (k,g),v in sorted(geo.items(), key=lambda kv:(kv[0][0],-len(kv[1]))):
            ngeo=sum(1 for (kk,_g) in geo if kk==k)
            if ngeo > 24: continue          # (one geometry per factorisation step: covered by the per-kernel totals above)
            if len(v)>=4 or k.startswith("syrk_tn_"): fh.write("%-32s %16s %7d %11.2f\n" % (k,g,len(v),sum(v)/len(v)))
    print(open(out+"/summary_full_bench.txt").read())
except Exception as e:
    print("full-bench summary failed:", e)
PY
rm -rf $OUT/trace68/*kernel_trace.csv
python - <<PY
import csv, glob, collections, re
out="$OUT"
def short(n):
    m=re.search(r'(\w+_kernel)', n); return m.group(1) if m else n.split("(")[0][:40]
with open(out+"/summary.txt","w") as fh:
    fh.write("== rocprofv3 --kernel-trace --stats : python bench.py --steps 20 --warmup 2 --no-cpu --train-rows 20000 --rcr68-shard 0 ==\n")
    fh.write("%-28s %7s %13s %11s %7s\n" % ("kernel","calls","total_us","avg_us","%"))
    for r in list(csv.DictReader(open(out+"/kernel_stats.csv")))[:14]:
        fh.write("%-28s %7s %13.1f %11.2f %7.2f\n" % (short(r["Name"]), r["Calls"], float(r["TotalDurationNs"])/1e3, float(r["AverageNs"])/1e3, float(r["Percentage"])))
    # the same kernels per launch geometry (training rows / detect batch are different problem sizes)
    geo=collections.defaultdict(list)
    for f in glob.glob(out+"/trace/*kernel_trace.csv"):
        for r in csv.DictReader(open(f)):
            k=short(r["Kernel_Name"])
            if k in ("hog_packed_kernel","desc_kernel","hog_fast_kernel","apply_partial_kernel","apply_tiled_kernel","apply_tiled_f16_kernel","apply_reduce_kernel","syrk_tn_kernel","syrk_tn_glds_kernel","syrk_tn_gldsw_kernel","syrk_tn_split_w4_kernel","syrk_update_f16_w4_kernel","syrk_tn_split_w8p_kernel","split_planes_f16_kernel"):
                geo[(k,"x".join(r.get(c,"?") for c in ("Grid_Size_X","Grid_Size_Y","Grid_Size_Z")))].append((float(r["End_Timestamp"])-float(r["Start_Timestamp"]))/1e3)
    fh.write("\n== kernel trace by launch geometry (avg over dispatches; the detect steps of bench.py are the rows with the most calls) ==\n")
    fh.write("%-28s %12s %7s %11s\n" % ("kernel","grid","calls","avg_us"))
    for (k,g),v in sorted(geo.items(), key=lambda kv:(kv[0][0],-len(kv[1]))):
        if len(v)>=4: fh.write("%-28s %12s %7d %11.2f\n" % (k,g,len(v),sum(v)/len(v)))
    rows=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
    for f in glob.glob(out+"/p*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k=short(r["Kernel_Name"])
            if k not in ("hog_packed_kernel","desc_kernel","hog_fast_kernel","apply_partial_kernel","apply_tiled_kernel","apply_tiled_f16_kernel","syrk_tn_kernel","syrk_tn_glds_kernel","syrk_tn_gldsw_kernel","syrk_tn_split_w4_kernel","syrk_update_f16_w4_kernel","syrk_tn_split_w8p_kernel","split_planes_f16_kernel"): continue
            k="%s grid=%s" % (k, r.get("Grid_Size","?"))
            rows[k][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[(k,r["Counter_Name"])]+=1
    # one kernel is launched with several problem sizes (training rows, detect batch): keep, per kernel, the launch
    # geometries of the timed detect steps -- the most frequent one and every other with at least a quarter of its
    # dispatches (the HOG kernel runs the levels with S <= 32 as landmark pairs, i.e. with half the grid) -- and
    # merge them into one per-launch average, which is what bench.py's HIP-event average is
    nmax=collections.Counter()
    for k in rows:
        base=k.split(" grid=")[0]; nmax[base]=max(nmax[base], max(cnt[(k,c)] for c in rows[k]))
    merged=collections.defaultdict(lambda: collections.defaultdict(float)); mcnt=collections.Counter(); geos=collections.defaultdict(list)
    for k,v in rows.items():
        base,g=k.split(" grid=")
        if max(cnt[(k,c)] for c in v) * 4 < nmax[base]: continue
        geos[base].append(g)
        for c,val in v.items():
            merged[base][c]+=val; mcnt[(base,c)]+=cnt[(k,c)]
    rows={"%s grid=%s" % (b, "+".join(sorted(geos[b], key=int, reverse=True))): v for b,v in merged.items()}
    cnt={("%s grid=%s" % (b, "+".join(sorted(geos[b], key=int, reverse=True))), c): n for (b,c),n in mcnt.items()}
    fh.write("\n== PMC (separate passes), per-dispatch averages over the detect launch geometries of each kernel ==\n")
    for k,v in sorted(rows.items()):
        fh.write(k+"\n")
        for c,val in sorted(v.items()):
            fh.write("   %-28s %14.6g  (n=%d)\n" % (c, val/cnt[(k,c)], cnt[(k,c)]))
        if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            fs=v["FETCH_SIZE"]/cnt[(k,"FETCH_SIZE")]; ws=v["WRITE_SIZE"]/cnt[(k,"WRITE_SIZE")]
            fh.write("   -> HBM traffic per launch: FETCH_SIZE %.1f KB x2 (gfx950 half-count correction, MI355X_MICROARCH.md HBM) + WRITE_SIZE %.1f KB = %.1f MB\n" % (fs, ws, (2*fs+ws)/1024))
import json
hbm={}
for k,v in rows.items():
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        hbm[k.split(" grid=")[0]]={"bytes_per_launch": (2*v["FETCH_SIZE"]/cnt[(k,"FETCH_SIZE")]+v["WRITE_SIZE"]/cnt[(k,"WRITE_SIZE")])*1024.0,
                                   "launch_geometry": k.split(" grid=")[1], "dispatches": cnt[(k,"FETCH_SIZE")]}
        for c in ("SQ_INSTS_VALU","SQ_INSTS_SALU","SQ_INSTS_LDS","SQ_WAVES","SQ_ACTIVE_INST_VALU","SQ_LDS_BANK_CONFLICT","SQ_LDS_IDX_ACTIVE","SQ_VALU_MFMA_BUSY_CYCLES","SQ_BUSY_CYCLES","GRBM_GUI_ACTIVE"):
            if c in v: hbm[k.split(" grid=")[0]][c]=v[c]/cnt[(k,c)]
json.dump({"batch": 4096, "source": "scripts/profile_bench.sh: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of python bench.py --steps 20 --warmup 2 --no-cpu --train-rows 20000 --rcr68-shard 0; FETCH_SIZE x2 (gfx950), KB units", "kernels": hbm}, open(out+"/hbm_traffic.json","w"), indent=1)
print(open(out+"/summary.txt").read())
PY
# the raw traces are tens of megabytes: only the summaries travel back (gpurun merges at most 64 MiB)
rm -rf $OUT/trace $OUT/trace68 $OUT/p1 $OUT/p2 $OUT/p3 $OUT/p4
