import csv,sys,glob
for f in sys.argv[1:]:
    rows=list(csv.DictReader(open(f)))
    tot=sum(float(r['TotalDurationNs']) for r in rows)
    print(f, 'total kernel ms', tot/1e6, 'launches', sum(int(r['Calls']) for r in rows))
    for r in rows[:22]:
        print('  %-44s %6s %8.1f us avg %6.2f%%'%(r['Name'].replace('(anonymous namespace)::','')[:44], r['Calls'], float(r['AverageNs'])/1e3, 100*float(r['TotalDurationNs'])/tot))
