"""Phase timeline of the bucketed apply's bucket blocks (diagnostic build: tools/build_variant.sh tl -DMEE_APPLY_TIMELINE=1; run with
MEE_LIB_PATH=build/libmeepo_hip_tl.so).  Stamps per block (100 MHz wall clock): 0 entry, 1 first round trip done, 2 entries fetched + keys in the
LDS table, 3 scans done, 4 thread 0's items done, 5 block done.   usage: apply_timeline.py [uniform|zipf] [located 0|1]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from meepoembedding_amd import LookupTable, OPT_ADAGRAD, synth, _lib
dev = torch.device("cuda", 0)
dist = sys.argv[1] if len(sys.argv) > 1 else "uniform"
located = int(sys.argv[2]) if len(sys.argv) > 2 else 1
keys_n, batch, dim = 100_000_000, 1 << 18, 64
t = LookupTable(int(keys_n / 0.75), dim, device=dev, max_batch=1 << 20, optimizer=OPT_ADAGRAD)
bench.populate(t, synth, keys_n, dim, dev, 1 << 20)
batches = bench.lookup_batches(synth, keys_n, batch, 8, dist, dev, seed=3)
grads = torch.randn(batch, dim, device=dev) * 0.01
out = torch.empty((batch, dim), device=dev); found = torch.empty(batch, dtype=torch.uint8, device=dev); slots = torch.empty(batch, dtype=torch.int64, device=dev)
for i in range(12):
    if located:
        t.find_located(batches[i % 8], out=out, found=found, slots=slots, prepare_apply=True)
        t.apply_adagrad(batches[i % 8], grads, lr=0.01, slots=slots)
    else:
        t.apply_adagrad(batches[i % 8], grads, lr=0.01)
    torch.cuda.synchronize()   # (the library sizes a batch's bucket count by what the latest FINISHED apply reported: let the host see it, as a training loop's does)
torch.cuda.synchronize()
n_blocks = 768
buf = np.zeros(16384 * 8 + 1024 * 128, dtype=np.uint64)
L = _lib.lib()
L.mee_debug_timeline.argtypes = [C.c_void_p, C.c_uint64]; L.mee_debug_timeline.restype = C.c_int
assert L.mee_debug_timeline(buf.ctypes.data, buf.size) == 0
spare = buf[16384 * 8:].reshape(1024, 128)
buf = buf[:n_blocks * 8]
tl = buf.reshape(n_blocks, 8)[:, :6].astype(np.float64) * 0.01   # us; bucket blocks only
t0 = tl[:, 0].min()
print(f"{dist}, located={located}: {tl.shape[0]} bucket blocks; block start relative to the first: median {np.median(tl[:, 0] - t0):.2f} us, p90 {np.percentile(tl[:, 0] - t0, 90):.2f}, max {(tl[:, 0] - t0).max():.2f}")
names = ["first round trip (selector, totals, run matrix)", "entries + keys into the LDS table", "scans + source sort (+ slot handles)", "work items (thread 0)", "rest of the block"]
for k in range(5):
    d = tl[:, k + 1] - tl[:, k]
    print(f"  {names[k]:52s} median {np.median(d):6.2f} us   p10 {np.percentile(d, 10):6.2f}   p90 {np.percentile(d, 90):6.2f}   max {d.max():6.2f}")
end = tl[:, 5] - t0
print(f"  block end relative to the first start: median {np.median(end):.2f} us, p90 {np.percentile(end, 90):.2f}, max {end.max():.2f}")

# ---- a skewed batch (skew_units): per block its units — slabs (with the merge the last slab of a bucket runs) and buckets
skew = [e for e in range(1024) if spare[e, 1] > 0]
if skew:
    us0 = float(min(spare[e, 0] for e in skew)) * 0.01
    us = lambda x: float(x) * 0.01 - us0 if x else float("nan")
    blocks = []
    for e in skew:
        n_sl, n_un = int(spare[e, 1]) & 0xffffffff, int(spare[e, 1]) >> 32
        slabs = []
        for i in range(min(n_sl, 6)):
            w = spare[e, 4 + 16 * i: 20 + 16 * i]
            info = int(w[5])
            slabs.append(dict(e=e, b=info & 0xffff, sub=(info >> 16) & 0xffff, size=info >> 32, found=us(w[0]), slab=us(w[1]), ticket=us(w[2]), coll=us(w[3]), merge=us(w[4]),
                              end=us(w[7]), R=int(w[6]) & 0xffffffff, ph=[us(x) for x in w[8:16]]))
        bks = []
        for i in range(min(n_un - n_sl, 6)):
            w = spare[e, 100 + 4 * i: 104 + 4 * i]
            bks.append(dict(start=us(w[0]), end=us(w[1]), b=int(w[2]) & 0xffffffff, size=int(w[2]) >> 32))
        blocks.append(dict(e=e, scan=us(spare[e, 0]), end=us(spare[e, 3]), slabs=slabs, buckets=bks, units=n_un, sched=us(spare[e, 2]), bar1=us(spare[e, 126])))
    end = np.array([b_['end'] for b_ in blocks]); units = np.array([b_['units'] for b_ in blocks])
    print(f"skewed batch: {len(blocks)} blocks, {units.sum()} units ({sum(len(b_['slabs']) for b_ in blocks)} slabs); totals scanned {np.median([b_['scan'] for b_ in blocks]):.2f} us (median) after the first block; "
          f"units per block median {np.median(units):.0f}, max {units.max()}; block end median {np.median(end):.2f} us, p90 {np.percentile(end, 90):.2f}, max {end.max():.2f}")
    print(f"  first unit known: median {np.nanmedian([b_['sched'] for b_ in blocks]):.2f} us, p90 {np.nanpercentile([b_['sched'] for b_ in blocks], 90):.2f}; slab blocks: known {np.nanmedian([b_['sched'] for b_ in blocks if b_['slabs']]):.2f}, past the first barrier {np.nanmedian([b_['bar1'] for b_ in blocks if b_['slabs']]):.2f}")
    rows = [r for b_ in blocks for r in b_['slabs']]
    f = lambda k: np.array([r[k] for r in rows])
    if rows:
        ph = np.array([r['ph'] for r in rows])
        print(f"  slabs: {len(rows)}; unit resolved {np.nanmedian(f('found')):.2f} us; slab pass median {np.nanmedian(f('slab') - f('found')):.2f} us, p90 {np.nanpercentile(f('slab') - f('found'), 90):.2f}, max {np.nanmax(f('slab') - f('found')):.2f}; "
              f"hand-off (drain + fence + ticket) median {np.nanmedian(f('ticket') - f('slab')):.2f}, max {np.nanmax(f('ticket') - f('slab')):.2f}; slab end median {np.nanmedian(f('ticket')):.2f}, max {np.nanmax(f('ticket')):.2f}")
        print("  inside the slab pass (medians): fetch + LDS table %.2f us, scans + source sort %.2f, work items (thread 0's wave) %.2f, barrier before the long runs %.2f, long runs %.2f" % (
            np.nanmedian(ph[:, 0] - f('found')), np.nanmedian(ph[:, 1] - ph[:, 0]), np.nanmedian(ph[:, 2] - ph[:, 1]), np.nanmedian(ph[:, 3] - ph[:, 2]), np.nanmedian(f('slab') - ph[:, 3])))
        mg = [r for r in rows if r['R'] > 0]
        if mg:
            g = lambda k: np.array([r[k] for r in mg]); mp = np.array([r['ph'] for r in mg])
            print(f"  merges: {len(mg)} buckets; records median {np.median(g('R')):.0f}, max {g('R').max()}; merge time (ticket -> end) median {np.nanmedian(g('end') - g('ticket')):.2f} us, max {np.nanmax(g('end') - g('ticket')):.2f}; merge end median {np.nanmedian(g('end')):.2f}, max {np.nanmax(g('end')):.2f}")
            print("  inside the (last) merge pass (medians): records collected %.2f us after the ticket, LDS table %.2f, scans %.2f, work items %.2f, rest %.2f" % (
                np.nanmedian(g('coll') - g('ticket')), np.nanmedian(mp[:, 4] - g('coll')), np.nanmedian(mp[:, 5] - mp[:, 4]), np.nanmedian(mp[:, 6] - mp[:, 5]), np.nanmedian(g('end') - mp[:, 6])))
        sizes = sorted({(r['b'], r['size']) for r in rows}, key=lambda x: -x[1])
        print(f"  split buckets: {len(sizes)}; sizes: " + " ".join(str(s_) for _, s_ in sizes[:12]) + " ...")
    bk_rows = [r for b_ in blocks for r in b_['buckets'] if r['size'] and r['size'] <= 1024]
    if bk_rows:
        d = np.array([r['end'] - r['start'] for r in bk_rows]); sz = np.array([r['size'] for r in bk_rows]); st = np.array([r['start'] for r in bk_rows])
        print(f"  bucket units: {len(bk_rows)}; size median {np.median(sz):.0f}, p90 {np.percentile(sz, 90):.0f}, max {sz.max()}; duration median {np.median(d):.2f} us, p90 {np.percentile(d, 90):.2f}, max {d.max():.2f}; start median {np.median(st):.2f}, p90 {np.percentile(st, 90):.2f}, max {st.max():.2f}; corr(duration, size) {np.corrcoef(d, sz)[0, 1]:.2f}")
    # wave 0's turns in the block's latest bucket unit: (time, kind 1 = quad, 2 = four tile items, 3 = done)
    durs = {1: [], 2: []}; nturn = []
    for e in skew:
        w = spare[e, 68:100].reshape(16, 2)[:, 0]
        ts = [(float(int(x) >> 8) * 0.01, int(x) & 3) for x in w if x]
        nturn.append(sum(1 for _, k in ts if k != 3))
        for (t_a, k_a), (t_b, _) in zip(ts, ts[1:]):
            if k_a in durs: durs[k_a].append(t_b - t_a)
    print(f"  wave 0's turns per bucket unit: median {np.median(nturn):.0f}, p90 {np.percentile(nturn, 90):.0f}; a turn of four tile items: median {np.median(durs[2]):.2f} us, p90 {np.percentile(durs[2], 90):.2f} ({len(durs[2])} turns); a quad turn: median {np.median(durs[1]) if durs[1] else float('nan'):.2f} us, p90 {np.percentile(durs[1], 90) if durs[1] else float('nan'):.2f} ({len(durs[1])} turns)")
    print("  the blocks that end last:")
    for b_ in sorted(blocks, key=lambda x: -x['end'])[:10]:
        parts = [f"slab(b{r['b']} size {r['size']} sub {r['sub']}: {r['found']:.1f}->{r['ticket']:.1f}" + (f", merge R={r['R']} ->{r['end']:.1f}" if r['R'] else "") + ")" for r in b_['slabs']]
        parts += [f"bucket(b{r['b']} size {r['size']}: {r['start']:.1f}->{r['end']:.1f})" for r in b_['buckets']]
        print(f"    block {b_['e']:4d} end {b_['end']:.1f} us: " + " ".join(parts))
    sys.exit(0)

raw = buf.reshape(n_blocks, 8)
bucket = (raw[:, 6] >> np.uint64(32)).astype(np.int64); size = (raw[:, 6] & np.uint64(0xffffffff)).astype(np.float64); xcc = (raw[:, 7] >> np.uint64(32)).astype(np.int64); hw = (raw[:, 7] & np.uint64(0xffffffff)).astype(np.int64)
cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 0x7    # HW_ID: wave 3:0, simd 5:4, pipe 7:6, cu 11:8, sh 12, se 15:13
items = tl[:, 4] - tl[:, 3]
print(f"  bucket sizes: median {np.median(size):.0f}, p10 {np.percentile(size, 10):.0f}, p90 {np.percentile(size, 90):.0f}; corr(items time, size) = {np.corrcoef(items, size)[0, 1]:.2f}; corr(items time, block index) = {np.corrcoef(items, np.arange(items.size))[0, 1]:.2f}")
print("  items time per position: median %.1f ns, p10 %.1f, p90 %.1f" % tuple(np.percentile(items / size * 1e3, [50, 10, 90])))
for x in sorted(set(xcc.tolist())):
    m = xcc == x
    print(f"  XCC {x}: {m.sum():4d} blocks, items median {np.median(items[m]):6.2f} us, end median {np.median(end[m]):6.2f}, end max {end[m].max():6.2f}")
key = xcc * 1000 + se * 100 + sh * 20 + cu
cnt = {k: int((key == k).sum()) for k in set(key.tolist())}
per = np.array([cnt[k] for k in key.tolist()])
for c in sorted(set(per.tolist())):
    m = per == c
    print(f"  blocks on a CU that holds {c} bucket blocks: {m.sum():4d}, items median {np.median(items[m]):6.2f} us, end median {np.median(end[m]):6.2f}")

for name, par in (("XCC parity", xcc & 1), ("bucket parity", bucket & 1)):
    print(f"  by {name}: even items median {np.median(items[par == 0]):6.2f} us, odd {np.median(items[par == 1]):6.2f} us")
q = np.argsort(bucket)
print("  items time by bucket index (tenths of the bucket range): " + " ".join(f"{np.median(items[q][i * len(q) // 10:(i + 1) * len(q) // 10]):.1f}" for i in range(10)))

for k in range(4):
    d = tl[:, k + 1] - tl[:, k]
    print(f"  phase {k} by XCC parity: even {np.median(d[(xcc & 1) == 0]):6.2f} us, odd {np.median(d[(xcc & 1) == 1]):6.2f} us")
