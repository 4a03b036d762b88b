import csv, glob, collections, json, sys
root = sys.argv[1]
out = {}
for f in glob.glob(root + '/*/*/*_counter_collection.csv'):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'vet::' in r['Kernel_Name'] and ('k_spatial' in r['Kernel_Name'] or 'k_transition' in r['Kernel_Name']):
            agg[(r['Kernel_Name'].split('(')[0][-40:], r['Counter_Name'])].append(float(r['Counter_Value']))
    for (k, c), v in agg.items():
        out.setdefault(k, {})[c] = sum(v) / len(v)
for f in glob.glob(root + '/trace/*/*_kernel_stats.csv'):
    for r in csv.DictReader(open(f)):
        if 'vet::' in r['Name']:
            out.setdefault('kernel_stats', {})[r['Name'].split('(')[0][-40:]] = {'calls': int(r['Calls']), 'avg_ns': float(r['AverageNs'])}
json.dump(out, open(root + '/summary.json', 'w'), indent=1)
print(json.dumps(out, indent=1))
