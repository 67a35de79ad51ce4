"""Locate entries on which k_project_bin_fast and k_project_bin_general disagree (FIXED64 maps are order-independent,
so any differing pixel points at a differing record)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle, slicer_amd
from slicer_amd import synth
BOX=1000.0; RND=dict(sgn=(-1,1,-1),face=3,center=(0.3,0.6,0.1),rcase=3.0)
LDS,LD2S=[3.0,3.25,3.5,3.75],[3.25,3.5,3.75,4.0]
n=1<<24; npix=4096
pos=synth.positions(0,n,BOX,seed=0x51CE2)
S=slicer_amd.Slicer(0,max_chunk=n)
d=S.to_device(pos)
def run():
    S.plane_begin(npix,0.25,LDS,LD2S,accum=slicer_amd.ACC_FIXED64,algo=slicer_amd.ALGO_BINNED,want_type_maps=False)
    S.file_begin([0,n,0,0,0,0],[0,0.0123,0,0,0,0],BOX,RND["sgn"],RND["face"],RND["center"],RND["rcase"])
    S.deposit_device(1,d,n); S.file_end()
    m=S.algo_mask()
    return [S.plane_read(p,want_types=False)[0] for p in range(4)], m
a,ma=run()
os.environ["SLICER_K1_GENERAL"]="1"
b,mb=run()
print("masks",hex(ma),hex(mb))
x,y,z=oracle.transform(pos,BOX,RND["sgn"],RND["face"],RND["center"],RND["rcase"])
for p in range(4):
    diff=np.argwhere(a[p]!=b[p])
    print("plane",p,"differing pixels",len(diff), diff[:12].tolist())
    if len(diff)==0: continue
    xs,ys,ms,idx=oracle.select_project(x,y,z,None,0.0123,LDS[p],LD2S[p],BOX,0,0.25,npix,want_index=True)
    gx=np.floor(xs.astype(np.float64)*npix).astype(int); gy=np.floor(ys.astype(np.float64)*npix).astype(int)
    cy,cx=diff[:,0].mean(),diff[:,1].mean()
    near=np.where((np.abs(gx-cx)<=2)&(np.abs(gy-cy)<=2))[0]
    for k in near:
        i=idx[k]
        X=np.longdouble(x[i])-np.longdouble(0.5); Y=np.longdouble(y[i])-np.longdouble(0.5); Z=np.longdouble(z[i])
        dd=np.sqrt(X*X+Y*Y+Z*Z); dec=np.arcsin(X/dd); ra=np.arctan2(Y,Z)
        sx=dec/np.longdouble(0.25)+np.longdouble(0.5); sy=ra/np.longdouble(0.25)+np.longdouble(0.5)
        def tie_dist(s):
            f=np.float32(s); lo=np.nextafter(f,np.float32(-1)); hi=np.nextafter(f,np.float32(2))
            t1=(np.longdouble(f)+np.longdouble(lo))/2; t2=(np.longdouble(f)+np.longdouble(hi))/2
            return float(min(abs(s-t1),abs(s-t2)))
        print(" cand particle",int(i),"raw",pos[i].tolist(),"xyz",float(x[i]),float(y[i]),float(z[i]),"xs,ys",xs[k].hex() if hasattr(xs[k],'hex') else float(xs[k]),float(ys[k]),"gx,gy",gx[k],gy[k],"tie dist sx %.3e sy %.3e"%(tie_dist(sx),tie_dist(sy)), "lim-|dec| %.3e lim-|ra| %.3e"%(float(0.25*(1+2/npix)*0.5-abs(dec)),float(0.25*(1+2/npix)*0.5-abs(ra))))
