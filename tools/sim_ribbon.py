"""Fluid model of mrw_lv_rows_balanced_kernel's schedule (profiles/r03_c4_shape.txt): 3 072 pieces of 890 transitions on 1 024
SIMDs, three per SIMD sharing its throughput, every piece = the head of one chain wave, then -- after its predecessor's
head -- the tail of another.  python tools/sim_ribbon.py [throughput of a SIMD with ONE active wave, default 0.6]"""
import numpy as np, sys
# processor-sharing simulation of the ribbon schedule: P pieces on S SIMDs (piece p on SIMD (p % 4) + 4*((p//4) % 256)), 3 per SIMD
G,T,S=2731,1000,1024
P=3072; L=890
def thr(n):  # SIMD throughput with n active waves (relative)
    return {0:0.0,1:float(sys.argv[1]) if len(sys.argv)>1 else 0.6,2:1.0,3:1.0}[min(n,3)]
pieces=[]
for p in range(P):
    r0=p*L; r1=min(r0+L,G*T)
    if r0>=r1: pieces.append(None); continue
    g0=r0//T; t0=r0-g0*T; tail_end=min(T,t0+(r1-r0)); head=(r1-r0)-(tail_end-t0)
    pieces.append(dict(g0=g0,t0=t0,tail=tail_end-t0,head=head,tail_end=tail_end))
simd=lambda p: (p%4)+4*((p//4)%256)
# state: phase 0 head remaining, then wait for flag (g0,t0), then tail remaining
state=[]
flag={}  # (g) -> published t
for p,pc in enumerate(pieces):
    if pc is None: state.append(None); continue
    state.append(dict(phase=0 if pc['head']>0 else 1, rem=pc['head'] if pc['head']>0 else pc['tail']))
t=0.0; done=0; n=sum(1 for x in state if x)
finish=np.zeros(P)
while done<n:
    # active pieces per simd
    act=[[] for _ in range(S)]
    for p,st in enumerate(state):
        if st is None or st.get('done'): continue
        pc=pieces[p]
        if st['phase']==1 and pc['t0']>0 and flag.get(pc['g0'],0)!=pc['t0']:
            continue  # waiting
        act[simd(p)].append(p)
    # time to next event
    dt=1e18
    for s in range(S):
        k=len(act[s])
        if k==0: continue
        rate=thr(k)/k
        for p in act[s]:
            dt=min(dt,state[p]['rem']/rate)
    if dt==1e18: print("deadlock"); break
    for s in range(S):
        k=len(act[s])
        if k==0: continue
        rate=thr(k)/k
        for p in act[s]:
            st=state[p]; st['rem']-=rate*dt
            if st['rem']<=1e-9:
                pc=pieces[p]
                if st['phase']==0:
                    if pc['head']<T: flag[pc['g0']+1]=pc['head']
                    st['phase']=1; st['rem']=pc['tail']
                    if pc['tail']==0: st['done']=True; done+=1; finish[p]=t+dt
                else:
                    if pc['tail_end']<T: flag[pc['g0']]=pc['tail_end']
                    st['done']=True; done+=1; finish[p]=t+dt
    t+=dt
print("makespan",t,"vs 3 full waves",3*T, "ratio %.3f"%(t/(3*T)), "ideal", G*T/S)
