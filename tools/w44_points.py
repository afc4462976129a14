"""Which interpolation points for Winograd F(4x4,3x3) in fp32?  (round 5; CPU, numpy)  Toom-Cook matrices for 5 finite points + infinity
(B^T solved from the bilinear identity, checked exact in float64), then the rounding error of one 4x4 output tile summed over 64
channels with every transform / product / sum in fp32 (and with the filter transform in float64): mean and max of |err| / max|y| over
random tiles (x ~ relu(N(0,1)), w ~ U(+-1/sqrt(9 Cin))).   python tools/w44_points.py > profiles/r05_wino44_points.txt"""
import numpy as np, itertools
from fractions import Fraction as Fr
np.random.seed(0)
def mats(pts):
    # Toom-Cook F(4,3) for 5 finite points + infinity; returns AT (4x6), G (6x3), BT (6x6) as float64 (exact rationals first)
    n=6; m=4; r=3
    a=[Fr(p) for p in pts]
    AT=[[ (a[j]**i if j<5 else (Fr(1) if i==m-1 else Fr(0))) for j in range(n)] for i in range(m)]
    G=[]
    for j in range(5):
        f=Fr(1)
        for k in range(5):
            if k!=j: f*= (a[j]-a[k])
        G.append([a[j]**k / f for k in range(r)])
    G.append([Fr(0),Fr(0),Fr(1)])
    # solve BT: for each l: sum_j AT[i][j] G[j][k] BT[j][l] = delta(l==i+k)
    ATf=np.array([[float(x) for x in row] for row in AT]); Gf=np.array([[float(x) for x in row] for row in G])
    M=np.zeros((m*r,n)); 
    BT=np.zeros((n,n))
    for l in range(n):
        rhs=np.zeros(m*r)
        for i in range(m):
            for k in range(r):
                M[i*r+k,:]=ATf[i,:]*Gf[:,k]
                rhs[i*r+k]=1.0 if l==i+k else 0.0
        sol,res,rk,sv=np.linalg.lstsq(M,rhs,rcond=None)
        BT[:,l]=sol
    return ATf,Gf,BT
def rescale(AT,G,BT):
    # scale rows of BT to have max |entry| = a 'nice' number and compensate in G (row j of G / s_j)
    return AT,G,BT
def conv_err(AT,G,BT,dt=np.float32,Udouble=False,Cin=64,trials=200):
    errs=[]
    for t in range(trials):
        d=np.maximum(np.random.randn(Cin,6,6),0)  # relu(N(0,1))
        g=(np.random.rand(Cin,3,3)*2-1)/np.sqrt(9*Cin)
        # exact
        y=np.zeros((4,4))
        for i in range(4):
            for j in range(4):
                y[i,j]=np.sum(d[:,i:i+3,j:j+3]*g)
        if Udouble:
            U=np.einsum('ai,cij,bj->cab',G,g,G).astype(dt)
        else:
            Gd=G.astype(dt); gd=g.astype(dt)
            U=np.einsum('ai,cij->caj',Gd,gd).astype(dt)
            U=np.einsum('caj,bj->cab',U,Gd).astype(dt)
        Bd=BT.astype(dt); dd=d.astype(dt)
        V=np.einsum('ai,cij->caj',Bd,dd).astype(dt)
        V=np.einsum('caj,bj->cab',V,Bd).astype(dt)
        Mm=(U*V).astype(dt)
        # accumulate over channels in fp32 sequentially-ish (pairwise via sum in float32)
        Ms=Mm.sum(0,dtype=dt)
        Ad=AT.astype(dt)
        Y=np.einsum('ia,ab->ib',Ad,Ms).astype(dt)
        Y=np.einsum('ib,jb->ij',Y,Ad).astype(dt)
        errs.append(np.abs(Y-y).max()/np.abs(y).max())
    return np.mean(errs), np.max(errs)
def direct_err(Cin=64,trials=200):
    errs=[]
    for t in range(trials):
        d=np.maximum(np.random.randn(Cin,6,6),0); g=(np.random.rand(Cin,3,3)*2-1)/np.sqrt(9*Cin)
        y=np.zeros((4,4)); y32=np.zeros((4,4),np.float32)
        for i in range(4):
            for j in range(4):
                y[i,j]=np.sum(d[:,i:i+3,j:j+3]*g)
                y32[i,j]=np.sum((d[:,i:i+3,j:j+3].astype(np.float32)*g.astype(np.float32)),dtype=np.float32)
        errs.append(np.abs(y32-y).max()/np.abs(y).max())
    return np.mean(errs), np.max(errs)
print('direct fp32', direct_err())
for name,pts in (('std 0,1,-1,2,-2',(0,1,-1,2,-2)),('0,1,-1,1/2,-1/2',(0,1,-1,Fr(1,2),Fr(-1,2))),('0,1,-1,1/2,-2',(0,1,-1,Fr(1,2),-2)),('0,1,-1,2,-1/2',(0,1,-1,2,Fr(-1,2))),('0,1/2,-1/2,3/2,-3/2',(0,Fr(1,2),Fr(-1,2),Fr(3,2),Fr(-3,2))),('0,1,-1,3/2,-3/2',(0,1,-1,Fr(3,2),Fr(-3,2)))):
    AT,G,BT=mats(pts)
    # sanity: exactness in float64
    e64=conv_err(AT,G,BT,dt=np.float64,trials=5)[1]
    print('%-22s exact-check %.1e | fp32 all: mean %.2e max %.2e | U in fp64: mean %.2e max %.2e | max|BT| %.2f max|AT| %.2f' % ((name,e64)+conv_err(AT,G,BT)+conv_err(AT,G,BT,Udouble=True)+(np.abs(BT).max(),np.abs(AT).max())))
print('--- symmetric scan (0, +-a, +-b)')
res=[]
for a,b in ((Fr(1,2),Fr(3,2)),(Fr(1,2),1),(Fr(1,2),2),(Fr(3,4),Fr(3,2)),(Fr(2,3),Fr(4,3)),(Fr(1,2),Fr(5,4)),(Fr(5,8),Fr(3,2)),(Fr(1,2),Fr(7,4)),(Fr(3,4),Fr(7,4)),(Fr(5,8),Fr(13,8)),(Fr(3,8),Fr(11,8)),(Fr(1,2),Fr(13,8)),(Fr(9,16),Fr(3,2))):
    AT,G,BT=mats((0,a,-a,b,-b))
    np.random.seed(1)
    m1=conv_err(AT,G,BT,trials=300)
    res.append((m1[0],a,b,m1[1],np.abs(BT).max(),np.abs(AT).max()))
for r in sorted(res): print('a=%s b=%s mean %.2e max %.2e |BT| %.2f |AT| %.2f' % (r[1],r[2],r[0],r[3],r[4],r[5]))
