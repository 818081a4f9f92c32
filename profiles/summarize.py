import csv,sys,glob
for d in sys.argv[1:]:
    f=glob.glob(d+'/**/*kernel_stats.csv',recursive=True)
    if not f: print(d,'no stats'); continue
    print('==',d)
    tot=0
    for r in csv.DictReader(open(f[0])):
        if 'rocclr' in r['Name'] or 'opt_reset' in r['Name']: continue
        print("  %-62s avg %8.1f ns  min %6s max %6s"%(r['Name'].replace('lmcma::(anonymous namespace)::','')[:62], float(r['AverageNs']), r['MinNs'], r['MaxNs']))
        tot+=float(r['AverageNs'])
    print('  sum of kernel averages per generation: %.1f us'%(tot/1e3))
