import ctypes as C

import numpy as np
import torch

from suo_slam_amd import _lib


def run_backbone_from_staged(net, xin_nhwc48):
    """xin: numpy [L,256,256,48] -> numpy logits [L,41,64,64] through suo_net_backbone."""
    x = torch.from_numpy(np.ascontiguousarray(xin_nhwc48, np.float32)).cuda()
    L = x.shape[0]
    out = torch.empty((L, 41, 64, 64), device="cuda")
    _lib.check(_lib.lib().suo_net_backbone(net._h, C.c_void_p(x.data_ptr()), L, C.c_void_p(out.data_ptr()),
                                           C.c_void_p(torch.cuda.current_stream().cuda_stream)), "suo_net_backbone")
    torch.cuda.synchronize()
    return out.cpu().numpy()
