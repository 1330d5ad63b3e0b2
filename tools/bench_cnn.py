"""Image-branch-only timing, NCHW vs channels_last (development aid)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn as nn
import torch.nn.functional as F
from geometric_aware_dense_matching_amd.cnn import PSPNet

torch.backends.cudnn.benchmark = True
B = 16
net = PSPNet().cuda().eval()


def fwd(x):
    f = net.feats
    y = f.maxpool(f.relu(f.bn1(f.conv1(x))))
    y = f.layer4(f.layer3(f.layer2(f.layer1(y))))
    y = net.psp(y)
    y = net.up_3(net.up_2(net.up_1(y)))
    return net.final(y)


for fmt in ("nchw", "nhwc"):
    x = torch.randn(B, 3, 256, 256, device="cuda")
    if fmt == "nhwc":
        net = net.to(memory_format=torch.channels_last)
        x = x.contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        try:
            for _ in range(3):
                fwd(x)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                fwd(x)
            torch.cuda.synchronize()
            print(fmt, "%.2f ms per batch of %d" % ((time.perf_counter() - t0) * 100, B))
        except Exception as e:
            print(fmt, "failed:", repr(e)[:200])
