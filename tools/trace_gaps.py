import csv, sys, collections, glob
d=sys.argv[1]
ks=list(csv.DictReader(open(glob.glob(d+'/*kernel_trace.csv')[0])))
ks.sort(key=lambda r:int(r['Start_Timestamp']))
cp=list(csv.DictReader(open(glob.glob(d+'/*memory_copy_trace.csv')[0])))
t_end=int(ks[-1]['End_Timestamp'])
span=float(sys.argv[2]) if len(sys.argv)>2 else 1.8e9
win0=t_end-int(span)
sel=[k for k in ks if int(k['End_Timestamp'])>win0]
iv=sorted((max(int(k['Start_Timestamp']),win0),int(k['End_Timestamp'])) for k in sel)
busy=0; cur_s,cur_e=iv[0]; gaps=[]
for s,e in iv[1:]:
    if s>cur_e:
        busy+=cur_e-cur_s; gaps.append((cur_e,s)); cur_s,cur_e=s,e
    else: cur_e=max(cur_e,e)
busy+=cur_e-cur_s
print("window ms",(t_end-win0)/1e6,"busy ms",busy/1e6,"idle ms",(t_end-win0-busy)/1e6)
big=[g for g in gaps if g[1]-g[0]>1e6]
print(len(big),"gaps >1ms, total",sum(b-a for a,b in big)/1e6)
for a,b in big[:60]:
    ov=[c for c in cp if int(c['Start_Timestamp'])<b and int(c['End_Timestamp'])>a]
    prev=[k for k in sel if int(k['End_Timestamp'])==a]
    nxt=[k for k in sel if int(k['Start_Timestamp'])==b]
    print(f"gap {(a-win0)/1e6:8.1f}..{(b-win0)/1e6:8.1f} ({(b-a)/1e6:6.2f} ms) copies:{[(c['Direction'][12:15],round((int(c['End_Timestamp'])-int(c['Start_Timestamp']))/1e6,1)) for c in ov]} prev {prev[0]['Kernel_Name'][:24] if prev else ''} q{prev[0]['Queue_Id'] if prev else ''} next {nxt[0]['Kernel_Name'][:24] if nxt else ''} q{nxt[0]['Queue_Id'] if nxt else ''}")
ev=[]
for k in sel:
    ev.append((max(int(k['Start_Timestamp']),win0),1,k['Queue_Id'])); ev.append((int(k['End_Timestamp']),-1,k['Queue_Id']))
ev.sort()
act=collections.Counter(); last=win0; hist=collections.Counter()
for t,dd,q in ev:
    n=sum(1 for v in act.values() if v>0)
    hist[n]+=t-last; last=t; act[q]+=dd
print("time by number of queues with a kernel in flight:",{k:round(v/1e6,1) for k,v in sorted(hist.items())})
big_k=[(int(k['End_Timestamp'])-int(k['Start_Timestamp']),k['Kernel_Name'][:30],k['Grid_Size_X']) for k in sel if 'copyBuffer' in k['Kernel_Name'] and int(k['End_Timestamp'])-int(k['Start_Timestamp'])>5e5]
print("copyBuffer kernels >0.5ms:",len(big_k),[ (round(a/1e6,1),g) for a,_,g in big_k][:20])
