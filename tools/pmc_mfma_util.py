"""MFMA-pipe utilisation per kernel from one rocprofv3 --pmc run (SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE):
busy cycles summed over the 1024 SIMDs / (1024 x active GPU cycles); GRBM_GUI_ACTIVE is summed over the 8 XCDs.

Two denominators are reported. `mfma_busy_frac` divides by GRBM_GUI_ACTIVE / 8, which MI355X_MICROARCH.md warns "reads
high on dispatches shorter than about 0.3 ms" (the quotient GUI_ACTIVE / duration comes out at 2.6 - 2.9 GHz for the
30 us LSTM launches, above the part's 2.4 GHz maximum), so it UNDER-states the busy share of short kernels.
`mfma_busy_frac_at_inkernel_clock` divides by dispatch duration x the in-kernel clock that the kernels' own
s_memtime / s_memrealtime stamps give under this load (2.17 GHz, tools/lstm_stamps.py): busy cycles per SIMD over the
cycles the SIMD actually had.

`gui_active_over_duration_ghz` is reported only for kernels whose mean dispatch is >= 100 us (and never above the part's
2.4 GHz): for 20 - 30 us launches GRBM_GUI_ACTIVE covers time outside the dispatch's start / end stamps and the quotient
reads 2.9 - 4.7 "GHz" (VERDICT r05 weak 6).

usage: python tools/pmc_mfma_util.py <rocprof output dir> [out.json] ["what was profiled": the program and its arguments]"""
INKERNEL_CLOCK_GHZ = 2.17
import csv, glob, collections, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from build_identity import build_identity
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
    util2 = a['busy'] / 1024 / (a['dur_us'] * INKERNEL_CLOCK_GHZ * 1e3)
    ghz = a['gui'] / a['dur_us'] / 1e3
    ghz_ok = a['dur_us'] / a['n'] >= 100.0 and ghz <= 2.4
    out[name] = {'dispatches': a['n'], 'mean_us': round(a['dur_us'] / a['n'], 1), 'gui_active_over_duration_ghz': round(ghz, 2) if ghz_ok else None,
                 'mfma_busy_frac': round(util, 3), 'mfma_busy_frac_at_inkernel_clock': round(util2, 3)}
    print('%-44s n %4d  mean %8.1f us  GUI_ACTIVE/duration %.2f GHz  MFMA pipe busy %5.1f %% (of GUI_ACTIVE)  %5.1f %% (of duration x %.2f GHz)' % (
        name, a['n'], a['dur_us'] / a['n'], a['gui'] / a['dur_us'] / 1e3, 100 * util, 100 * util2, INKERNEL_CLOCK_GHZ))
if len(sys.argv) > 2:
    what = sys.argv[3] if len(sys.argv) > 3 else 'python3 tools/probe_engine.py fp32 512 threestep'
    json.dump({'source': 'rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -- %s (every launch on one stream)' % what,
               'definitions': {'mfma_busy_frac': 'SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / (GRBM_GUI_ACTIVE / 8 XCDs): share of the cycles the counter saw; '
                                                 'under-states short dispatches (GUI_ACTIVE covers time outside them)',
                               'mfma_busy_frac_at_inkernel_clock': 'the same busy cycles / (dispatch duration x %.2f GHz, the clock in-kernel stamps give under this '
                                                                   'load): share of the cycles a SIMD had during the dispatch; under-states kernels the chip clocks '
                                                                   'below that (the bf16 MFMA kernels: 1.9 - 2.0 GHz)' % INKERNEL_CLOCK_GHZ,
                               'gui_active_over_duration_ghz': 'effective clock, only for mean dispatches >= 100 us and <= 2.4 GHz; null otherwise'},
               'kernels': out, **build_identity()}, open(sys.argv[2], 'w'), indent=1)
