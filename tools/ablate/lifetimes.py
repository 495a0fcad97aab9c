import json,sys
for ln in sys.stdin:
    try: d=json.loads(ln)
    except Exception: continue
    if 'lifetimes_us' in d:
        L=d['lifetimes_us']; ends=sorted(e for _,_,e in L); lifes=sorted(l for _,l,_ in L)
        n=len(L); q=lambda a,f: a[min(n-1,int(f*n))]
        print('n',n,'life deciles',[round(q(lifes,f/10),1) for f in range(0,11)])
        print('end deciles',[round(q(ends,f/10),1) for f in range(0,11)])
        # by blockIdx%8 (XCD)
        import collections
        x=collections.defaultdict(list)
        for i,l,e in L: x[i%8].append(e)
        print('mean end by xcd',{k:round(sum(v)/len(v),1) for k,v in sorted(x.items())})
    elif 'us' in d: print('us',d['us'],'TOPS',d['TOPS'])
