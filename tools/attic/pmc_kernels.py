"""Per-kernel SQ counter summary of one rocprofv3 --pmc run (last dispatch of every kernel/grid)."""
import csv, glob, collections, sys
f = (glob.glob(sys.argv[1] + '/*/*counter_collection.csv') + glob.glob(sys.argv[1] + '/*counter_collection.csv'))[0]
disp = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    if 'ds::' not in r['Kernel_Name']: continue
    d = disp.setdefault(int(r['Dispatch_Id']), {'name': r['Kernel_Name'].split('(')[0].replace('void ds::', '').replace('ds::', ''),
                                                 'grid': int(r['Grid_Size']) // int(r['Workgroup_Size']),
                                                 'dur': (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3})
    d[r['Counter_Name']] = d.get(r['Counter_Name'], 0) + float(r['Counter_Value'])
last = collections.OrderedDict()
for k in disp.values(): last[(k['name'], k['grid'])] = k
for (name, grid), k in last.items():
    wc = max(k.get('SQ_WAVE_CYCLES', 0), 1)
    extra = ' '.join('%s=%.3g' % (c.replace('SQ_', ''), v) for c, v in k.items() if c.startswith('SQ_') and c not in ('SQ_WAVE_CYCLES', 'SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY'))
    print('%-44s wgs %6d dur %8.1fus wait_any %3.0f%% wait_inst %3.0f%% active %3.0f%%  %s' % (
        name, grid, k['dur'], 100 * k.get('SQ_WAIT_ANY', 0) / wc, 100 * k.get('SQ_WAIT_INST_ANY', 0) / wc, 100 * k.get('SQ_ACTIVE_INST_ANY', 0) / wc, extra))
