cd /tmp && export TMPDIR=/tmp
export ISE_NO_DIRECT=1
OUT=$GRAFT_REPO_ROOT/gpurun_out/bimodal; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -- python3 $GRAFT_REPO_ROOT/scripts/merge_bimodal_probe.py > $OUT/log.txt 2>&1
grep " us " $OUT/log.txt
python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# find searches of the 6000-row indexes: short_scan launches with small grids
sel = [r for r in rows if "short_scan" in r["Kernel_Name"] and int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]) < 250]
print("short launches on small indexes:", len(sel))
# group consecutive launches into 10 indexes of 350 each
per = 350
idx = {id(r): i for i, r in enumerate(rows)}
for g in range(len(sel) // per):
    chunk = sel[g * per + 50:(g + 1) * per]
    durs = collections.defaultdict(list)
    gaps = []
    for r in chunk:
        i = idx[id(r)]
        durs["short"].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
        for j in (1, 2):
            rr = rows[i + j]
            nm = "merge" if "merge" in rr["Kernel_Name"] else ("gate" if "exact_scan" in rr["Kernel_Name"] else rr["Kernel_Name"][:20])
            durs[nm].append(int(rr["End_Timestamp"]) - int(rr["Start_Timestamp"]))
        gaps.append(int(rows[i + 1]["Start_Timestamp"]) - int(r["End_Timestamp"]))
        gaps.append(int(rows[i + 2]["Start_Timestamp"]) - int(rows[i + 1]["End_Timestamp"]))
    print(g, {k: round(sum(v) / len(v) / 1000, 2) for k, v in durs.items()}, "gap", round(sum(gaps) / len(gaps) / 1000, 2))
PY
