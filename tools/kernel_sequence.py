"""the dispatches of a rocprofv3 --kernel-trace run IN ORDER, one token per dispatch (short kernel name + duration in us), for the kernels whose names contain one of
the given substrings — what each of the first skewed steps of a stream launched and how long each launch ran (tools/first_skewed_batch.py under the profiler).
usage: kernel_sequence.py DIR [first-index [count]] [name-substring ...]"""
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
args = sys.argv[2:]
first = int(args.pop(0)) if args and args[0].lstrip("-").isdigit() else 0
count = int(args.pop(0)) if args and args[0].isdigit() else 1 << 30
pats = args or ["find_prepare", "bkt_apply", "bkt_sort", "find_kernel"]
rows = [r for r in csv.DictReader(open(f)) if any(p in r["Kernel_Name"] for p in pats)]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))


def short(n):
    m = re.search(r"(\w+)<([^>]*)>", n)
    return (m.group(1) + "<" + m.group(2).replace(" ", "") + ">") if m else n.split("(")[0][-40:]


if first < 0:
    first = max(0, len(rows) + first)
t_prev = None
for i, r in enumerate(rows[first:first + count], first):
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = "" if t_prev is None else f" (+{(s - t_prev) / 1e3:.0f} idle)"
    print(f"{i:5d} {short(r['Kernel_Name']):60s} {(e - s) / 1e3:8.1f} us{gap}  grid {r.get('Grid_Size', '?')} wg {r.get('Workgroup_Size', '?')}")
    t_prev = e
