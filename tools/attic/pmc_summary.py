import csv, glob, collections, sys
f = glob.glob(sys.argv[1] + '/*/*counter_collection.csv')[0]
rows = list(csv.DictReader(open(f)))
disp = collections.OrderedDict()
for r in rows:
    if 'ds::' not in r['Kernel_Name']: continue
    d = disp.setdefault(int(r['Dispatch_Id']), {'name': r['Kernel_Name'].split('(')[0].replace('void ds::','').replace('ds::',''), 'grid': int(r['Grid_Size'])//int(r['Workgroup_Size']),
        'dur': (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3})
    d[r['Counter_Name']] = d.get(r['Counter_Name'], 0) + float(r['Counter_Value'])
ks = list(disp.values())
idx = [i for i,k in enumerate(ks) if 'stem1' in k['name']]
step = ks[idx[-1]:]
seen = set()
for k in step:
    key = (k['name'], k['grid'])
    if key in seen: continue
    seen.add(key)
    wc = max(k.get('SQ_WAVE_CYCLES',0),1)
    gui = k.get('GRBM_GUI_ACTIVE',0)/8
    print('%-34s wgs %5d dur %7.1fus clk %.2fGHz wait_any %3.0f%% wait_inst %3.0f%% active %3.0f%% mfma_busy/SIMD %8.0f (%.0f%% of gui) ldsconf %8.0f' % (
        k['name'], k['grid'], k['dur'], gui/k['dur']/1e3 if k['dur'] else 0, 100*k.get('SQ_WAIT_ANY',0)/wc, 100*k.get('SQ_WAIT_INST_ANY',0)/wc, 100*k.get('SQ_ACTIVE_INST_ANY',0)/wc,
        k.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/1024, 100*k.get('SQ_VALU_MFMA_BUSY_CYCLES',0)/1024/max(gui,1), k.get('SQ_LDS_BANK_CONFLICT',0)))
