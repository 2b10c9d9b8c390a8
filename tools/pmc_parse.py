import csv, collections, sys
rows = list(csv.DictReader(open(sys.argv[1])))
pat = sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for r in rows:
    if pat in r['Kernel_Name']:
        agg[r['Dispatch_Id']][r['Counter_Name']] += float(r['Counter_Value'])
        agg[r['Dispatch_Id']]['dur_us'] = (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
d = list(agg.values())[-1]
print({k: (round(v, 1) if k == 'dur_us' else int(v)) for k, v in d.items()})
