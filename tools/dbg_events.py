import sys, os
ROOT='/root/repo'
sys.path[:0]=[ROOT, os.path.join(ROOT,'tracking-anything-with-deva_amd')]
import torch, bench
from workload import synth
torch.set_grad_enabled(False)
dev=torch.device('cuda:0')
net,_=bench.build_network(dev, split=True)
cfg=synth.base_config(enable_long_term=False, enable_long_term_count_usage=False)
frames=bench.make_clip(480,854,46,seed=100,device=dev)
core=bench.start_clip(net,cfg,frames,5,dev)
for t in range(1,6): core.step(frames[t])
torch.cuda.synchronize()
more=bench.make_clip(480,854,22,seed=999,device=dev)
for f in more[:2]: core.step(f)
torch.cuda.synchronize()
import time
with bench.ConvTimer() as ct:
    marks=[]
    for f in more[2:]:
        marks.append(len(ct.records)); core.step(f)
rows=[(r[1].elapsed_time(r[2]), i, r[3], r[5]) for i,r in enumerate(ct.records)]
rows.sort(reverse=True)
for ms,i,sig,cls in rows[:15]:
    fr=max(j for j,m in enumerate(marks) if m<=i)
    print('%.3f ms  record %d (frame %d, #%d in frame) sig %s class %d'%(ms,i,fr,i-marks[fr],sig,cls))
print('total', sum(r[0] for r in rows), 'frames', len(marks))
