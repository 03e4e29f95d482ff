cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/tr -- python $GRAFT_REPO_ROOT/tools/step_times.py 3 > $GRAFT_REPO_ROOT/gpurun_out/tr.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/tr -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
# last 6 groups worth: print from the last "k_conv_act" 4 launches back
idx = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("void k_conv_act") or "k_conv_act" in r["Kernel_Name"]]
start = idx[-3] if len(idx) >= 3 else 0
prev_end = None
for r in rows[start:]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    gap = (s - prev_end) / 1e6 if prev_end is not None else 0.0
    print("%10.3f ms  dur %8.3f ms  gap %7.3f  %s" % (s / 1e6, (e - s) / 1e6, gap, r["Kernel_Name"][:60]))
    prev_end = max(prev_end or 0, e)
PY
tail -3 gpurun_out/tr.log
