"""Time qg_policy_mid_head_sample (middle layer + head + draw) on random activations.  Run on the GPU box."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qiskit_gym_amd.collector import mid_head_sample, pack_head, pack_mid

for B, A in ((65536, 170), (65536, 214), (8192, 170)):
    g = torch.Generator(device="cpu").manual_seed(1)
    h = torch.randn((B, 512), generator=g).relu().to(torch.bfloat16).cuda()
    w2 = (torch.randn((256, 512), generator=g) * 0.05).to(torch.bfloat16).cuda()
    b2 = torch.randn(256, generator=g).to(torch.bfloat16).cuda()
    wh = (torch.randn((A + 1, 256), generator=g) * 0.05).to(torch.bfloat16).cuda()
    bh = torch.randn(A + 1, generator=g).to(torch.bfloat16).cuda()
    pm, ph = pack_mid(w2, b2), pack_head(wh, bh, A, A, after_mid=True)
    outs = mid_head_sample(h, pm, 256, ph, A, 1, 0)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(20):
        mid_head_sample(h, pm, 256, ph, A, 1, i, actions=outs[0], logp=outs[1], entropy=outs[2], values=outs[3])
    b.record()
    torch.cuda.synchronize()
    print(f"mid_head_sample {B} envs, {A} actions: {a.elapsed_time(b) / 20 * 1e3:.1f} us", flush=True)
