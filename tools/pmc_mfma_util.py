"""MFMA-pipe utilisation per kernel from one rocprofv3 --pmc run (SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE):
busy cycles summed over the 1024 SIMDs / (1024 x active GPU cycles); GRBM_GUI_ACTIVE is summed over the 8 XCDs."""
import csv, glob, collections, json, sys
f = (glob.glob(sys.argv[1] + '/*/*counter_collection.csv') + glob.glob(sys.argv[1] + '/*counter_collection.csv'))[0]
disp = collections.OrderedDict()
for r in csv.DictReader(open(f)):
    if 'ds::' not in r['Kernel_Name']: continue
    d = disp.setdefault(int(r['Dispatch_Id']), {'name': r['Kernel_Name'].split('(')[0].replace('void ds::', '').replace('ds::', ''),
                                                 'grid': int(r['Grid_Size']) // int(r['Workgroup_Size']),
                                                 'dur_us': (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3})
    d[r['Counter_Name']] = d.get(r['Counter_Name'], 0) + float(r['Counter_Value'])
agg = collections.OrderedDict()
for k in disp.values():
    a = agg.setdefault(k['name'], {'n': 0, 'busy': 0.0, 'gui': 0.0, 'dur_us': 0.0})
    a['n'] += 1; a['busy'] += k.get('SQ_VALU_MFMA_BUSY_CYCLES', 0); a['gui'] += k.get('GRBM_GUI_ACTIVE', 0) / 8; a['dur_us'] += k['dur_us']
out = {}
for name, a in agg.items():
    if a['gui'] <= 0: continue
    util = a['busy'] / 1024 / a['gui']
    out[name] = {'dispatches': a['n'], 'mean_us': round(a['dur_us'] / a['n'], 1), 'clock_ghz': round(a['gui'] / a['dur_us'] / 1e3, 2), 'mfma_busy_frac': round(util, 3)}
    print('%-44s n %4d  mean %8.1f us  clk %.2f GHz  MFMA pipe busy %5.1f %%' % (name, a['n'], a['dur_us'] / a['n'], a['gui'] / a['dur_us'] / 1e3, 100 * util))
if len(sys.argv) > 2:
    json.dump({'source': 'rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -- python3 tools/probe_engine.py fp32 512 (every launch on one stream)', 'kernels': out}, open(sys.argv[2], 'w'), indent=1)
