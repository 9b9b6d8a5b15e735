# Prototype of the projection solvers' safeguard: accelerated proximal gradient (FISTA, gradient restart) on the conic dual;
# run from the repo root after collecting failing instances with tests/soak/network_fuzz.py-style loops.
import pickle, numpy as np, sys
fails=pickle.load(open('/tmp/fails.pkl','rb'))
A_PERS=(1/60)*(208/1000)*5
def setup(net,a,dem):
    n=net.num_stations
    ph=np.deg2rad(net.phase_angles)
    Bre=net.constraint_matrix*np.cos(ph)[None,:]; Bim=net.constraint_matrix*np.sin(ph)[None,:]
    b=a.astype(np.float64)*32; h=np.minimum(dem.astype(np.float64)/A_PERS/32,1.0)*32
    return Bre,Bim,net.magnitudes.copy(),b,h
def resid(Bre,Bim,r,b,h,z):
    nu=Bre.T@z[:,0]+Bim.T@z[:,1]; y=np.clip(b-nu,0,h)
    w=np.stack([Bre@y,Bim@y],1); nz=np.hypot(z[:,0],z[:,1]); nw=np.hypot(w[:,0],w[:,1])
    act=nz>0
    ra=np.where(act, np.hypot(*(w-r[:,None]*z/np.maximum(nz,1e-300)[:,None]).T)/r, 0).max()
    ri=np.where(~act, nw/r-1, -1).max()
    return y,w,ra,ri
def fista(Bre,Bim,r,b,h,iters=200000,tol=1e-12,z0=None):
    m=len(r); B=np.vstack([Bre,Bim]); L=np.linalg.eigvalsh(B@B.T).max(); t=1.0/L
    z=np.zeros((m,2)) if z0 is None else z0.copy(); v=z.copy(); th=1.0
    for k in range(iters):
        nu=Bre.T@v[:,0]+Bim.T@v[:,1]; y=np.clip(b-nu,0,h)
        g=np.stack([Bre@y,Bim@y],1)
        u=v+t*g; nu_=np.hypot(u[:,0],u[:,1])
        zn=u*np.maximum(0,1-t*r/np.maximum(nu_,1e-300))[:,None]
        # gradient restart
        if ((zn-z)*(v-zn)).sum()>0: th=1.0; vn=zn.copy()
        else:
            thn=(1+np.sqrt(1+4*th*th))/2; vn=zn+(th-1)/thn*(zn-z); th=thn
        z,v=zn,vn
        if k%25==0:
            y,w,ra,ri=resid(Bre,Bim,r,b,h,z)
            if ra<=tol and ri<=1e-10: return z,y,k,ra,ri
    y,w,ra,ri=resid(Bre,Bim,r,b,h,z)
    return z,y,iters,ra,ri
for f in fails:
    case,t,e,net,a,dem,kkt=f
    S=setup(net,a,dem)
    z,y,k,ra,ri=fista(*S)
    print(case,t,e,'n',net.num_stations,'m',len(S[2]),'iters',k,'res',ra,ri,'active',int((np.hypot(z[:,0],z[:,1])>0).sum()))

sys.path.insert(0,'/root/repo/tests')
from helpers import random_network
from oracle import binding as ob
from sustaingym_amd.network import caltech_acn, jpl_acn
rng=np.random.default_rng(5)
nets=[caltech_acn(), jpl_acn()]+[random_network(rng,f'r{i}') for i in range(40)]
its=[]; bad=0; worst=0; disagree=0
for ni,net in enumerate(nets):
    onet=ob.OracleNetwork(net); n=net.num_stations
    for trial in range(25):
        occ=rng.random(n)<rng.uniform(0.3,1.0)
        dem=np.where(occ, rng.uniform(0.05,40,n),0).astype(np.float32)
        a=np.where(rng.random(n)<0.5,1.0,rng.random(n)) if trial%2 else rng.random(n)**0.3
        S=setup(net,a,dem)
        z,y,k,ra,ri=fista(*S,iters=20000)
        ok = ra<=1e-9 and ri<=1e-10
        bad += (not ok); its.append(k)
        x,rc,kkt=onet.project(a,dem)
        if rc==0:
            d=np.abs(x*32-y).max(); worst=max(worst,d)
            if d>1e-4: disagree+=1
print('instances',len(its),'fista failures',bad,'iters median',np.median(its),'p90',np.percentile(its,90),'max',max(its),'max |y_newton-y_fista|',worst,'disagree',disagree)
