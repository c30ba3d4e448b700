"""A CPU model of the run numbering of `k_runs_wave<true>` (forma_amd/csrc/paint.hip: runs numbered per tile row, no counting
pass), statement by statement — the per-lane break / head / row-start masks, the per-wave summaries in LDS, the tile's status
word (aggregate, or at once the next index of its last row), the look-back over the predecessors' words, the per-slot index
of a sweep — checked against the definition: a run head's index = the segment index where its tile row begins + the number
of heads of that row in front of it; row_base[row] = where the row begins.  Guards the index arithmetic (several rows inside
one 2 048-segment tile, chunks of more than 64 runs, rows of more than 64 tiles, unpaintable rows and columns, a partial last
tile); the kernel itself is compared with the oracle by tests/test_gpu_round2.py::test_runs_numbered_per_tile_row_without_a_counting_pass."""
import random

import pytest

RW_SEGS=8; RW_CHUNK=512; RN_TILE=2048; WAVES=4
def clz64(x): return 64 - x.bit_length()
def popc(x): return bin(x).count('1')
def run(seed, n, tiles_w, tiles_h, nrows_span, maxrun, randomize_status=True):
    rnd=random.Random(seed)
    # sorted stream of hi words
    keys=[]
    while len(keys) < n // max(1, (maxrun + 1) // 2) + 1:
        keys.append((rnd.randrange(0,nrows_span), rnd.randrange(0,tiles_w+3), rnd.randrange(0,6)))
    # make runs: repeat keys
    stream=[]
    for k in sorted(keys):
        stream += [k]*rnd.randrange(1,maxrun+1)
        if len(stream)>=n: break
    stream=stream[:n]; n=len(stream)
    hi=[(t<<21)|(x<<9)|l for (t,x,l) in stream]
    def paintable(h): tyb=h>>21; txb=(h>>9)&0xFFF; return (tyb-1)%(1<<32) < tiles_h and txb<=tiles_w
    # definition
    exp_j={}; exp_base={}
    row_start=None; cnt=0
    for i in range(n):
        brk = i==0 or hi[i]!=hi[i-1]
        rs = i==0 or (hi[i]>>21)!=(hi[i-1]>>21)
        if rs: row_start=i; cnt=0; 
        if rs and ((hi[i]>>21)-1)%(1<<32) < tiles_h: exp_base[(hi[i]>>21)-1]=i
        if brk and paintable(hi[i]): exp_j[i]=row_start+cnt; cnt+=1
    ntiles=(n+RN_TILE-1)//RN_TILE
    status=[0]*ntiles  # (flag,value)
    got_j={}; got_base={}
    def lookback(tb):
        excl=0; p0=tb-1
        while p0>=0:
            vals=[]
            for lane in range(64):
                p=p0-lane
                vals.append(status[p] if p>=0 else (2,0))
            pref=[l for l in range(64) if vals[l][0]==2]
            fp=pref[0] if pref else 64
            for l in range(min(fp,63)+1):
                assert vals[l][0]!=0, "would spin forever in the sequential model"
            excl+=sum(vals[l][1] for l in range(64) if l<=fp)
            if fp<64: break
            p0-=64
        return excl
    for tb in range(ntiles):
        ws=[]; lanes=[]
        for w in range(WAVES):
            cbase=tb*RN_TILE+w*RW_CHUNK
            chunk_n=min(RW_CHUNK,n-cbase) if cbase<n else 0
            L=[]
            for lane in range(64):
                g0=cbase+lane*RW_SEGS
                h=[]
                for q in range(RW_SEGS):
                    i=g0+q
                    if chunk_n and lane*RW_SEGS+q>=chunk_n: h.append(hi[cbase+chunk_n-1])
                    elif i<n: h.append(hi[i])
                    else: h.append(hi[(i-g0)%n])   # garbage (stream head)
                L.append(h)
            # phi
            ph=[None]*64
            for lane in range(64):
                ph[lane]=L[lane-1][RW_SEGS-1] if lane>0 else 0
            if cbase>0 and cbase<n: ph[0]=hi[cbase-1]
            else: ph[0]=(~L[0][0])&0xFFFFFFFF if cbase==0 else 0
            bm=[0]*64; hm=[0]*64; rm=[0]*64
            for lane in range(64):
                q_hi=ph[lane]
                for q in range(RW_SEGS):
                    if L[lane][q]!=q_hi: bm[lane]|=1<<q
                    q_hi=L[lane][q]
                if chunk_n:
                    q_hi=ph[lane]
                    for q in range(RW_SEGS):
                        if (bm[lane]>>q)&1:
                            tybq=L[lane][q]>>21; txbq=(L[lane][q]>>9)&0xFFF
                            if (tybq-1)%(1<<32)<tiles_h and txbq<=tiles_w: hm[lane]|=1<<q
                            if (q_hi>>21)!=tybq: rm[lane]|=1<<q
                        q_hi=L[lane][q]
            nh=[popc(x) for x in hm]; tot=sum(nh)
            rsl=[l for l in range(64) if rm[l]]
            aft=tot; rsp1=0
            if rsl:
                Ls=rsl[-1]; ql=rm[Ls].bit_length()-1
                aft=sum(nh[l] for l in range(Ls+1,64))+popc(hm[Ls]>>ql)
                rsp1=cbase+Ls*RW_SEGS+ql+1
            ws.append((tot,aft,rsp1)); lanes.append((cbase,chunk_n,L,ph,bm))
        wt=[x[0] for x in ws]; wa=[x[1] for x in ws]; wr=[x[2] for x in ws]
        ul=-1
        for v in range(WAVES):
            if wr[v]: ul=v
        val=0; flag=1
        for v in range(WAVES):
            if v==ul: val=(wr[v]-1)+wa[v]; flag=2
            elif v>ul: val+=wt[v]
        status[tb]=(flag,val)
        final_prefix=None
        for w in range(WAVES):
            cbase,chunk_n,L,ph,bm=lanes[w]
            u=-1
            for v in range(WAVES):
                if v<w and wr[v]: u=v
            if u>=0:
                jrow=0
                for v in range(WAVES):
                    if v==u: jrow=(wr[v]-1)+wa[v]
                    elif v>u and v<w: jrow+=wt[v]
            lb_pending=False; pfx_pending=False; pfx_add=0
            if u<0:
                # the row continues from the tiles in front: looked up where the first index is needed (below)
                lb_pending=True
                jrow=sum(wt[v] for v in range(w))
                if w==WAVES-1 and ul<0: pfx_pending=True; pfx_add=jrow+wt[WAVES-1]
            if not chunk_n: continue
            # slots
            brk=[lane*RW_SEGS+q for lane in range(64) for q in range(RW_SEGS) if (bm[lane]>>q)&1]
            R=len(brk)
            first_hi=L[0][0]; before_hi=ph[0]
            prev_tile=0
            s0=0
            while s0<=R:
                tile=[0]*64; st=[0]*64; mine=[False]*64; tyb=[0]*64; valv=[False]*64
                for lane in range(64):
                    sg=s0+lane
                    mine[lane]= 1<=sg<=R
                    if mine[lane]:
                        st[lane]=brk[sg-1]; kh=hi[cbase+st[lane]]
                        tile[lane]=kh>>9; tyb[lane]=kh>>21; txb=(kh>>9)&0xFFF
                        valv[lane]=(tyb[lane]-1)%(1<<32)<tiles_h and txb<=tiles_w
                bv=sum(1<<l for l in range(64) if valv[l])
                rsb=0; pos=[cbase+st[l] for l in range(64)]
                for lane in range(64):
                    sg=s0+lane
                    pt=tile[lane-1] if lane>0 else prev_tile
                    if sg==1: pt=(first_hi if st[lane] else before_hi)>>9
                    if mine[lane] and (((pt>>12)!=tyb[lane]) or pos[lane]==0): rsb|=1<<lane
                prev_tile=tile[63]
                jabs=[False]*64; jpart=[0]*64
                for lane in range(64):
                    below=(1<<lane)-1
                    rs_le=rsb&(((below<<1)|1)&((1<<64)-1))
                    jabs[lane]=rs_le!=0
                    if rs_le:
                        r=rs_le.bit_length()-1
                        jpart[lane]=pos[r]+popc(bv&below&~((1<<r)-1))
                    else: jpart[lane]=popc(bv&below)
                    if (rsb>>lane)&1 and (tyb[lane]-1)%(1<<32)<tiles_h: got_base[tyb[lane]-1]=pos[lane]
                if lb_pending and (any(valv[l] and not jabs[l] for l in range(64)) or pfx_pending):
                    lb=lookback(tb)
                    jrow+=lb; lb_pending=False
                    if pfx_pending: final_prefix=(2,lb+pfx_add)
                    pfx_pending=False
                for lane in range(64):
                    if valv[lane]:
                        assert jabs[lane] or not lb_pending
                        got_j[pos[lane]]=jpart[lane] if jabs[lane] else jrow+jpart[lane]
                if rsb:
                    rl=rsb.bit_length()-1; jrow=pos[rl]+popc(bv>>rl); lb_pending=False
                else: jrow+=popc(bv)
                s0+=64
        if final_prefix and (not randomize_status or rnd.random()<0.5): status[tb]=final_prefix
    assert got_j==exp_j, (seed, [ (k,got_j.get(k),exp_j.get(k)) for k in sorted(set(got_j)|set(exp_j)) if got_j.get(k)!=exp_j.get(k)][:5])
    assert got_base==exp_base, (seed, got_base, exp_base)
    return len(exp_j), ntiles


@pytest.mark.parametrize("seed", range(40))
def test_chain_numbering_equals_the_definition(seed):
    rnd = random.Random(1000 + seed)
    n = rnd.choice([1, 7, 511, 512, 513, 2047, 2048, 2049, 5000, 9000, 20000])
    tiles_w = rnd.choice([1, 3, 8, 40]); tiles_h = rnd.choice([1, 2, 5, 30])
    span = rnd.choice([1, 2, tiles_h + 2, tiles_h + 3, 3 * tiles_h + 2])
    maxrun = rnd.choice([1, 2, 5, 40, 700, 5000])
    run(seed, n, tiles_w, tiles_h, span, maxrun)


def test_a_row_of_more_than_64_tiles_takes_several_probes():
    heads, ntiles = run(4, 300000, 4000, 1, 2, 3)
    assert ntiles > 128 and heads > 10000
