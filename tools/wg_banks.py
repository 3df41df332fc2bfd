"""Bank-conflict search for the raw-row B image of the LDS-DMA weight gradient (WgImgB2 / WgImgBG in csrc/gz_igemm.h):
for every geometry and chunk rectangle, the row / channel pitches (in quads of padding) that minimise the worst number of
distinct addresses on one bank among the 32 (c, ky, kx) lanes of a half-wave, over all alignments of a 32-column block."""
import itertools, math
def analyze(KH,KW,S,P,CW):
    R=16//CW; TAPS=KH*KW
    LP=(P+3)//4*4
    maxcol=S*(CW-1)+KW-1-P+LP
    RQ0=maxcol//4+1
    rows=S*(R-1)+KH
    best=None
    for pr in range(0,4):
        RP=(RQ0+pr)*4
        for pc in range(0,9):
            CHP=rows*RP+4*pc
            worst=0; tot=0; cnt=0
            for n0 in range(0, TAPS*32, 32):   # all alignments of a 32-col block
                banks={}
                for l in range(32):
                    n=n0+l; c=n//TAPS; tap=n%TAPS; ky=tap//KW; kx=tap%KW
                    a=c*CHP+ky*RP+kx
                    banks.setdefault(a%32,set()).add(a)
                m=max(len(v) for v in banks.values())
                worst=max(worst,m); tot+=m; cnt+=1
            key=(worst, tot/cnt, CHP)
            if best is None or key<best[0]: best=(key,RP,CHP,pr,pc)
    return dict(KH=KH,S=S,CW=CW,rows=rows,RQ0=RQ0,best=best)
for (KH,KW,S,P) in [(5,5,2,2),(3,3,1,1),(4,4,2,1)]:
    for CW in (16,8,4):
        print(analyze(KH,KW,S,P,CW))
