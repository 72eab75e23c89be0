"""First policy layer, CliffordGym 16q x B envs, hidden 512: the bit-consuming MFMA kernel (qg_vec_embed)
against what it replaces (bf16 observation expansion + hipBLASLt GEMM with bias/ReLU epilogue)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from qiskit_gym_amd.collector import embed, pack_embedding
from qiskit_gym_amd.vec import VecEnv
from util import line_gateset


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


for B in (int(x) for x in os.environ.get("BATCHES", "8192,65536,262144").split(",")):
    gs = line_gateset("clifford", 16)
    env = VecEnv("clifford", 16, gs, B, add_inverts=False, add_perms=False, track_solution=False, difficulty=256)
    env.reset(1)
    H = 512
    w = (torch.randn((H, 1024), device="cuda") * 0.05).to(torch.bfloat16)
    bias = torch.randn(H, device="cuda")
    packed = pack_embedding(env, w)
    out = torch.empty((B, H), dtype=torch.bfloat16, device="cuda")
    x = torch.empty((B, 1024), dtype=torch.bfloat16, device="cuda")
    bb = bias.to(torch.bfloat16)
    t_bits = timeit(lambda: embed(env, packed, bias, H, relu=True, out=out))
    t_obs = timeit(lambda: env.observe_as(torch.bfloat16, out=x))
    t_gemm = timeit(lambda: torch._addmm_activation(bb, x, w.t()))
    t_pack = timeit(lambda: pack_embedding(env, w, out=packed))
    flops = 2.0 * B * 1024 * H
    print(f"B={B}: embed from bits {t_bits:.1f} us ({flops / t_bits * 1e-6:.0f} TFLOP/s) | observe bf16 {t_obs:.1f} us + GEMM {t_gemm:.1f} us "
          f"({flops / t_gemm * 1e-6:.0f} TFLOP/s) = {t_obs + t_gemm:.1f} us | weight repack {t_pack:.1f} us", flush=True)
    ref = torch._addmm_activation(bb, x, w.t()).float()
    print(f"   max |bits - gemm| = {(out.float() - ref).abs().max().item():.4f} (bf16 bias in the GEMM path)")
