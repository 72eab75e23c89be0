"""qg_policy_mid_head_sample at growing batches (isolated launches): the slope is the steady-state cost of a 32-env tile per wave, the
intercept what filling and draining the pipeline costs.  Run on the GPU box.  B list on the command line."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qiskit_gym_amd.collector import mid_head_sample, pack_head, pack_mid

A = 170
for B in [int(x) for x in sys.argv[1:]] or [32768, 65536, 131072, 262144]:
    g = torch.Generator(device="cpu").manual_seed(1)
    h = torch.randn((B, 512), generator=g).relu().to(torch.bfloat16).cuda()
    w2 = (torch.randn((256, 512), generator=g) * 0.05).to(torch.bfloat16).cuda()
    b2 = torch.randn(256, generator=g).to(torch.bfloat16).cuda()
    wh = (torch.randn((A + 1, 256), generator=g) * 0.05).to(torch.bfloat16).cuda()
    bh = torch.randn(A + 1, generator=g).to(torch.bfloat16).cuda()
    pm, ph = pack_mid(w2, b2), pack_head(wh, bh, A, A, after_mid=True)
    outs = mid_head_sample(h, pm, 256, ph, A, 1, 0)
    torch.cuda.synchronize()
    pad = torch.empty(16 << 20, dtype=torch.float32, device="cuda")
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    pad.fill_(1.0)
    torch.cuda.synchronize()
    a.record()
    for i in range(20):
        pad.fill_(float(i))
    b.record()
    torch.cuda.synchronize()
    fill_us = a.elapsed_time(b) / 20 * 1e3
    a.record()
    for i in range(20):
        mid_head_sample(h, pm, 256, ph, A, 1, i, actions=outs[0], logp=outs[1], entropy=outs[2], values=outs[3])
        pad.fill_(float(i))
    b.record()
    torch.cuda.synchronize()
    print(f"mid_head_sample {B} envs, {A} actions (isolated launches): {a.elapsed_time(b) / 20 * 1e3 - fill_us:.1f} us", flush=True)
