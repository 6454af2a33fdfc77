import sys, numpy as np, torch
sys.path.insert(0,'tests'); sys.path.insert(0,'oracle'); sys.path.insert(0,'.')
import oracle_np as o
from multiview_motion_capture_amd import device as dev
g=np.load('tests/golden/ik_converged.npz'); d=torch.device('cuda:0')
n,VP=g['poses'].shape[:2]
kps=torch.from_numpy(np.ascontiguousarray(g['poses'].reshape(1,n*VP,1,17,3))).to(d)
Pm=np.ascontiguousarray(g['projs'].reshape(n*VP,3,4)).copy(); Pm[np.abs(Pm).sum(axis=(1,2))==0]=np.eye(3,4)
mem=-np.ones((n,VP),dtype=np.int32)
for i in range(n): mem[i,:g['n_views'][i]]=i*VP+np.arange(g['n_views'][i])
p,j,info=dev.ik_solve_stages(torch.from_numpy(g['init']).to(d),3,int(g['max_nfev']),kps17=kps,Pmats=torch.from_numpy(Pm).to(d),members=torch.from_numpy(mem).to(d))
torch.cuda.synchronize(); p,j,info=p.cpu().numpy(),j.cpu().numpy(),info.cpu().numpy()
both=(g['s1_status']>0)&(g['s2_status']>0)
rel=(info[:,3]-g['s2_cost'])/g['s2_cost']
def observed(poses,v,mv):
    sc=np.array([o.add_mid_spine(q) for q in poses[:v]])[:,o.IK_OBS_IDX,2]
    return o.IK_SKEL_IDX[(sc>0.1).sum(axis=0)>=mv]
rows=[]
for i in np.nonzero(both)[0]:
    v=int(g['n_views'][i]); oj=observed(g['poses'][i],v,2)
    dj=np.abs(j[i][oj]-g['joints'][i][oj]).max()/np.abs(g['joints'][i]).max() if len(oj) else np.nan
    rows.append((i,v,int(g['source'][i]),int(g['warm_init'][i]),rel[i],dj,info[i,1],info[i,4],g['s1_nfev'][i],g['s2_nfev'][i]))
rows=np.array(rows)
np.save('gpurun_out/conv_rows.npy',rows)
for v in (2,3,4,5):
    m=rows[:,1]==v
    same=m&(np.abs(rows[:,4])<1e-6)
    print('views',v,'n',m.sum(),'same-min',same.sum(),'dj quantiles 50/90/99/max', np.nanquantile(rows[same,5],[.5,.9,.99,1.]) if same.sum() else None, '| rel<-1e-4 (device better):',(rows[m,4]<-1e-4).sum())
bad=rows[(np.abs(rows[:,4])<1e-6)&(rows[:,5]>1e-4)]
print('offenders (same min, dj>1e-4):'); print(bad[:,[0,1,2,3,4,5]])
print('device nfev mean', rows[:,6].mean(), rows[:,7].mean(), 'ref nfev mean', rows[:,8].mean(), rows[:,9].mean())
