import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows = [r for r in rows if 'zkw_' in r['Kernel_Name']]
t0 = min(int(r['Start_Timestamp']) for r in rows)
# take the last 40% of cycle kernels as steady state
cyc = sorted([r for r in rows if 'cycle_kernel' in r['Kernel_Name']], key=lambda r: int(r['Start_Timestamp']))
print("n cycle kernels", len(cyc), "queues", len(set(r['Queue_Id'] for r in cyc)))
tail = cyc[len(cyc)//2:]
span = int(tail[-1]['End_Timestamp']) - int(tail[0]['Start_Timestamp'])
busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in tail)
print("steady: %d kernels over %.3f ms; mean dur %.3f ms; mean concurrency %.2f; %.3f ms/step" % (len(tail), span/1e6, busy/len(tail)/1e6, busy/span, span/len(tail)/1e6))
for r in tail[:24]:
    print(r['Queue_Id'], r['Kernel_Name'][:20], (int(r['Start_Timestamp'])-t0)/1e3, (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
# per-kernel gaps within a queue
byq = collections.defaultdict(list)
for r in sorted(rows, key=lambda r: int(r['Start_Timestamp'])): byq[r['Queue_Id']].append(r)
q = list(byq)[0]
print("queue", q)
prev = None
for r in byq[q][-15:]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    print("  %-28s start %.1f us dur %.1f us gap %.1f us" % (r['Kernel_Name'][:28], (s-t0)/1e3, (e-s)/1e3, (s-prev)/1e3 if prev else 0))
    prev = e
# full timeline of a 2.5 ms steady-state window
w0 = int(tail[8]['Start_Timestamp'])
print("window:")
for r in sorted(rows, key=lambda r: int(r['Start_Timestamp'])):
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if w0 <= s < w0 + 2500000:
        print("  q%-3s %-22s start %8.1f dur %7.1f  grid %s wg %s lds %s" % (r['Queue_Id'], r['Kernel_Name'][4:26], (s - w0)/1e3, (e - s)/1e3, r.get('Grid_Size_X', r.get('Grid_Size')), r.get('Workgroup_Size_X', r.get('Workgroup_Size')), r.get('LDS_Block_Size')))
