import sys, numpy as np
sys.path.insert(0,'/root/repo')
from raytracer_amd import sppm
W,H,M=1920,1080,1000000
ctx=sppm.SppmContext(sppm.make_config(W,H,photons_per_pass=M))
ctx.eye_pass(0)
s=ctx.hitpoints(); order=ctx.tile_order()
pos=s['hp_f'][:,1:4]; valid=s['valid']>0
print('n', len(order), 'unique', len(np.unique(order)), 'valid', valid.sum())
po=pos[order]; vo=valid[order]
nt=len(order)//64
pt=po[:nt*64].reshape(nt,64,3); vt=vo[:nt*64].reshape(nt,64)
big=1e30
lo=np.where(vt[...,None],pt,big).min(1); hi=np.where(vt[...,None],pt,-big).max(1)
ext=(hi-lo).max(1); live=vt.any(1)
print('tile extent percentiles', np.percentile(ext[live],[10,50,90,99,99.9,100]))
for k in (16749,16750,16754, 100, 20000):
    print(k, lo[k].round(2), hi[k].round(2), vt[k].sum())
    print(pt[k][:6].round(3))
