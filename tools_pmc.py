import csv,glob,collections,sys
f=glob.glob(sys.argv[1]+'/*/*counter_collection.csv')[0]
agg=collections.defaultdict(lambda: collections.defaultdict(float))
cnt=collections.Counter()
for r in csv.DictReader(open(f)):
    agg[r['Kernel_Name'][:70]][r['Counter_Name']]+=float(r['Counter_Value'])
for k,v in agg.items():
    if any(x in k for x in sys.argv[2:]): print(k, {a:f'{b:.4g}' for a,b in v.items()})
