"""print mee:: kernels from a rocprofv3 --kernel-trace --stats output dir"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'mee::' in r['Name'] or (len(sys.argv) > 2 and sys.argv[2] in r['Name']):
        print(r['Name'].split('(')[0][-48:].ljust(50), r['Calls'].rjust(6), 'avg %8.1f us' % (float(r['AverageNs']) / 1e3), 'min %8.1f max %8.1f' % (float(r['MinNs']) / 1e3, float(r['MaxNs']) / 1e3))
