import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from gan_lab_amd import ops
def timeit(fn, reps=5):
    fn(); fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
B=32
for cin, cout, h in [(16,16,1024),(16,32,1024),(32,32,512),(64,64,256)]:
    x = torch.randn(B, cin, h, h, device='cuda'); w = torch.randn(cout, cin, 3, 3, device='cuda')
    g = ops.Geom(B, cin, h, h, cout, 3, 1, 0)
    fl = 2.0*9*cin*cout*h*h*B
    out=[]
    for dbg in (0,1,2,3,4,5,6,7):
        os.environ['GANLAB_CONV_DBG']=str(dbg)
        t = timeit(lambda: ops.k_conv_fwd(x, w, None, g, 0.05))
        out.append(f'dbg{dbg}:{t:.3f}ms')
    os.environ['GANLAB_CONV_DBG']='0'
    print(f'{cin}->{cout}@{h}:', ' '.join(out), f'(ideal MFMA {fl/157.3e9:.3f}ms)')
