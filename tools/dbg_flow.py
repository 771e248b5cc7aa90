import sys; sys.path.insert(0,'.')
import numpy as np, torch
from oracle import oracle
from rmnet_amd import ops
g=np.load('tests/golden/flow_affine.npz')
f32=np.float32
for n in sorted({k.split('.')[0] for k in g.files}):
    flow,m1,m2,want=g[n+'.flow'],g[n+'.m1'],g[n+'.m2'],g[n+'.out']
    got=ops.update_optical_flow(flow,m1,m2)
    bad=np.argwhere(got.view(np.uint32)!=want.view(np.uint32))
    print(n, len(bad))
    for (i,j,c) in bad[:4]:
        fj,fi=f32(j),f32(i)
        a=m2.ravel(); b=m1.ravel()
        x2r=f32(f32(f32(a[0]*fj)+f32(a[1]*fi))+a[2]); y2r=f32(f32(f32(a[3]*fj)+f32(a[4]*fi))+a[5])
        x1=f32(fj+flow[i,j,0]); y1=f32(fi+flow[i,j,1])
        x1r=f32(f32(f32(b[0]*x1)+f32(b[1]*y1))+b[2])
        print('  ',(i,j,c),'got',got[i,j],'want',want[i,j],'x2raw',repr(x2r),'y2raw',repr(y2r),'x1raw',repr(x1r), 'flow', flow[i,j])
