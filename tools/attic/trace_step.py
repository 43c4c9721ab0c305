import csv, glob, sys
f = glob.glob(sys.argv[1] + '/*/*_kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
rows = [r for r in rows if 'ds::' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'stem1' in r['Kernel_Name']]
s, e = idx[-2], idx[-1]
step = rows[s:e]
t0 = int(step[0]['Start_Timestamp'])
prev_end = t0
for r in step:
    name = r['Kernel_Name'].split('(')[0].replace('void ds::', '').replace('ds::', '')
    st, en = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    wg = int(r['Grid_Size_X']) // max(int(r['Workgroup_Size_X']), 1)
    print('%-28s wgs %6d  start %8.1f dur %7.1f us  gap %6.1f us  vgpr %s agpr %s lds %s' % (
        name, wg, (st - t0) / 1e3, (en - st) / 1e3, (st - prev_end) / 1e3, r.get('VGPR_Count', ''), r.get('Accum_VGPR_Count', ''), r.get('LDS_Block_Size', '')))
    prev_end = max(prev_end, en)
print('step total %.1f us' % ((max(int(r['End_Timestamp']) for r in step) - t0) / 1e3))
