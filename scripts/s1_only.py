import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from densematchingbenchmark_amd import ops
dev = torch.device("cuda:0")
x = torch.randn(4, 32, 48, 136, 240, device=dev); w = torch.randn(32, 32, 3, 3, 3, device=dev) * 0.03
wp = ops.pack_conv3d_weights(w)
for _ in range(6): y = ops.conv3d_k3(x, wp, 32)
torch.cuda.synchronize()
