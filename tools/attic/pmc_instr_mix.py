import csv, glob, collections, sys
rows = []
for d in sys.argv[1:]:
    f = glob.glob(d + '/*/*counter_collection.csv')[0]
    rows += [dict(r, _src=d) for r in csv.DictReader(open(f))]
agg = collections.OrderedDict()
for r in rows:
    if 'ds::' not in r['Kernel_Name']: continue
    name = r['Kernel_Name'].split('(')[0].replace('void ds::','').replace('ds::','')
    key = (name, int(r['Grid_Size'])//int(r['Workgroup_Size']))
    d = agg.setdefault(key, collections.defaultdict(float))
    d[r['Counter_Name']] += float(r['Counter_Value'])
    d['_n_' + r['Counter_Name']] += 1
for key, d in agg.items():
    if not ('gemm' in key[0] or 'fused' in key[0]): continue
    print(key)
    out = []
    for c in sorted(k for k in d if not k.startswith('_n_')):
        n = d['_n_' + c]   # dispatches x (xcc/se dims)
        out.append('%s=%.3g' % (c.replace('SQ_',''), d[c]))
    print('   ' + '  '.join(out))
