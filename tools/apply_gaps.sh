#!/bin/bash
# usage (GPU box): tools/apply_gaps.sh [uniform|zipf] — kernel timeline of the apply-alone loop (tools/apply_host_cost.py): gaps between its launches
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf /tmp/agp; timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/agp -o x -- python3 tools/apply_host_cost.py $1 2>&1 | grep enqueue
python3 - <<'PY'
import csv, glob
f = glob.glob('/tmp/agp/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if 'bkt_' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
rows = rows[-400:]
import statistics
gap_sa, gap_as, d_s, d_a = [], [], [], []
for a, b in zip(rows, rows[1:]):
    g = (int(b['Start_Timestamp']) - int(a['End_Timestamp'])) / 1e3
    if 'sort' in a['Kernel_Name'] and 'apply' in b['Kernel_Name']: gap_sa.append(g)
    if 'apply' in a['Kernel_Name'] and 'sort' in b['Kernel_Name']: gap_as.append(g)
for r in rows:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    (d_s if 'sort' in r['Kernel_Name'] else d_a).append(d)
print(f"sort {statistics.median(d_s):.1f} us, gap sort->apply {statistics.median(gap_sa):.2f} us, apply {statistics.median(d_a):.1f} us, gap apply->sort {statistics.median(gap_as):.2f} us; "
      f"period {(int(rows[-1]['Start_Timestamp']) - int(rows[0]['Start_Timestamp'])) / 1e3 / (len(rows) - 1) * 2:.1f} us per step")
PY
