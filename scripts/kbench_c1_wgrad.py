import torch, sys, os
sys.path.insert(0, os.getcwd())
from densematchingbenchmark_amd import ops
dev=torch.device("cuda:0")
x=torch.randn(4,32,48,64,128,device=dev); dc=torch.randn(4,1,48,64,128,device=dev)
for _ in range(5): ops.conv3d_k3_wgrad(x,dc)
torch.cuda.synchronize()
s,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(50): ops.conv3d_k3_wgrad(x,dc)
e.record(); torch.cuda.synchronize()
us=s.elapsed_time(e)/50*1e3
print("head wgrad [4,32,48,64,128]: %.1f us  (x read once: %.2f TB/s)"%(us, x.numel()*4/us/1e6))
