"""Development aid: how many 64 KiB tile slots a recycled colour-plane ring would need.  Replays the plan's
processing order (column strips, boustrophedon, 8 chunks, rim first) with `rows` persistent workgroups per XCD, one patch per
period; a slot lives from its patch's store phase until the tile's last contributor has stored, plus the sum latency and `slack`
periods; greedy interval colouring gives the slot count.
    python scripts/sim/ring_slots.py"""
import numpy as np, sys, heapq
def order_for(nli, nlj, n_xcd=8):
    cell = -np.ones((nli, nlj), int); idx = 0
    coords=[]
    # calculate_covering order irrelevant: use cell ids
    cell = np.arange(nli*nlj).reshape(nli,nlj)
    n = nli*nlj
    strips = 4 if nlj >= 8 else 1
    order=[]
    for s in range(strips):
        ja, jb = nlj*s//strips, nlj*(s+1)//strips
        for step in range(nli):
            li = nli-1-step if s & 1 else step
            for lj in range(ja, jb): order.append(cell[li,lj])
    chunk = (n+7)//8
    def rim(i):
        li, lj = divmod(i, nlj); return li in (0, nli-1) or lj in (0, nlj-1)
    out=[]
    for x in range(8):
        seg = order[x*chunk:(x+1)*chunk]
        out += [i for i in seg if rim(i)] + [i for i in seg if not rim(i)]
    return out, chunk
def sim(nli, nlj, rows, slack, sumlat=0.3):
    order, chunk = order_for(nli, nlj)
    n = nli*nlj
    t_end = np.zeros(n)
    for s, i in enumerate(order):
        x = s % chunk
        t_end[i] = x/rows + 1.0   # in patch periods
    nti, ntj = nli+1, nlj+1
    ivs=[]
    for ti in range(nti):
        for tj in range(ntj):
            who=[(ti-a)*nlj+(tj-b) for a in (0,1) for b in (0,1) if 0<=ti-a<nli and 0<=tj-b<nlj]
            death = max(t_end[w] for w in who) + sumlat + slack
            for w in who: ivs.append((t_end[w]-0.25, death))  # written during the store phase at the end of the patch
    ivs.sort()
    free=[]; live=[]; nslots=0; peak=0
    for a,b in ivs:
        while live and live[0][0] <= a:
            heapq.heappop(live); free.append(1)
        if free: free.pop()
        else: nslots+=1
        heapq.heappush(live,(b,))
        peak=max(peak,len(live))
    return nslots, len(ivs)
for (nli,nlj,rows) in ((33,33,31),(65,65,28)):
    for slack in (0, 0.5, 1, 2):
        ns, tot = sim(nli,nlj,rows,slack)
        print(f"lattice {nli}x{nlj} rows/XCD {rows} slack {slack}: slots {ns} of {tot} = {ns*64/1024:.0f} MB of {tot*64/1024:.0f} MB")
