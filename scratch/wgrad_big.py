import sys
src = open('/root/repo/scratch/wgrad_micro.py').read()
src = src[:src.index("case(64, 64, 32, 32, 3, splits=(384")]
exec(src)
import torch
from hipnet import _capi as C
# correctness of the config against the default one
def grads(N, H, Cin, Cout):
    torch.manual_seed(0)
    x = torch.randn(N, H, H, Cin, device='cuda').to(dt); dy = torch.randn(N, H, H, Cout, device='cuda').to(dt)
    ns = C.call('hrnet_wgrad_splits', 1, N, H, H, Cout, Cin, 1, 1)
    slabs = torch.empty(ns, Cout, 1, Cin, device='cuda'); g = torch.zeros(Cout, Cin, 1, 1, device='cuda')
    C.call('hrnet_conv2d_wgrad', 1, x.data_ptr(), dy.data_ptr(), None, None, slabs.data_ptr(), N, H, H, Cin, H, H, Cout, 1, 1, 0, ns, C.stream_ptr())
    C.call('hrnet_wgrad_reduce', slabs.data_ptr(), g.data_ptr(), ns, Cout, Cin, 1, Cout, Cin, 0, 0, C.stream_ptr())
    ref = torch.einsum('nhwo,nhwi->oi', dy.float(), x.float())
    return (g[:, :, 0, 0] - ref).abs().max().item() / ref.abs().max().item(), ns
print('rel err, ns', grads(4, 32, 480, 480), grads(2, 16, 256, 256))
case(64, 64, 480, 480, 1, splits=(None, 4, 8, 16))
case(64, 64, 64, 256, 1, splits=(None, 64, 128))
case(64, 64, 256, 256, 1, splits=(None, 16, 32))
