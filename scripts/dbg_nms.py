import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, ROOT)
import numpy as np, torch, ssd_amd
from oracle import ops
from test_gpu_stages import synth_heads, dev
rng = np.random.default_rng(1)
anc = ops.anchors(128,128); N = anc.shape[0]
synth_heads(rng,2,N,80)
synth_heads(rng,2,N,80,frac=0.01); rng.uniform(-1.0,4.0,N)
codes, logits = synth_heads(rng,1,N,80)
logits[0,100:140,3] = 2.0
rb, rl, rs, rn = ops.postprocess(logits, codes, anc, 0.15, 0.6, 25)
print("oracle", rn, np.bincount(rl[0][:rn[0]], minlength=80)[:8])
for fm in ["512","0"]:
    os.environ["SSD_NMS_FAST_MAX"]=fm
    b,s,c,n = ssd_amd.batch_multiclass_non_max_suppression(dev(torch,codes), dev(torch,anc), dev(torch,logits), 0.15,0.6,25)
    c=c.cpu().numpy(); n=n.cpu().numpy()
    print("gpu fast_max",fm, n, np.bincount(c[0][:n[0]], minlength=80)[:8])
    k=[i for i in range(n[0]) if c[0][i]==3]
    print(" class3 scores", s.cpu().numpy()[0][k][:6])
logits[0,100:140,3] = 2.0 + np.arange(40)*0.01
rb, rl, rs, rn = ops.postprocess(logits, codes, anc, 0.15, 0.6, 25)
print("untied oracle", rn)
for fm in ["512","0"]:
    os.environ["SSD_NMS_FAST_MAX"]=fm
    b,s,c,n = ssd_amd.batch_multiclass_non_max_suppression(dev(torch,codes), dev(torch,anc), dev(torch,logits), 0.15,0.6,25)
    print("untied gpu",fm,n.cpu().numpy())
