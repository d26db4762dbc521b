"""print the kernel timeline (start offset, duration, stream/queue) of a window of a rocprofv3 kernel trace csv"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows = [r for r in rows if 'mee::' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
lo, hi = int(sys.argv[2]), int(sys.argv[3])
t0 = int(rows[lo]['Start_Timestamp'])
for r in rows[lo:hi]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:7.1f} us  q{r.get('Queue_Id', '?'):>3}  {r['Kernel_Name'].split('(')[0][-60:]}")
