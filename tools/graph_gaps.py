import csv, sys, glob
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if 'mdie' in r['Kernel_Name']]
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows)
# split into steps at conv_first_kernel
starts = [i for i, e in enumerate(ev) if 'conv_first' in e[2]]
res = []
for a, b in zip(starts[-12:-1], starts[-11:]):
    seg = ev[a:b]
    t0, t1 = seg[0][0], ev[b][0]
    # union busy
    busy = 0; cur_s, cur_e = seg[0][0], seg[0][1]
    for s, e, _ in seg[1:]:
        if s > cur_e: busy += cur_e - cur_s; cur_s, cur_e = s, e
        else: cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    res.append((t1 - t0, busy, sum(e - s for s, e, _ in seg)))
import statistics as st
print("step span us: %.1f  union-busy us: %.1f  idle us: %.1f  sum-kernel us: %.1f" % (st.median(r[0] for r in res)/1e3, st.median(r[1] for r in res)/1e3, st.median(r[0]-r[1] for r in res)/1e3, st.median(r[2] for r in res)/1e3))
