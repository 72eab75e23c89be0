"""Time qg_policy_embed_words against what it replaces (packed rows -> bf16 observation -> library GEMM with bias + ReLU).
Run on the GPU box: python tools/bench_embed_words.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from qiskit_gym_amd.collector import _linear_relu, embed_words, expand_packed, pack_embed_words  # noqa: E402


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


SHAPES = (("PauliGym 20q (40 x 45)", 40, 45, 65536, 512), ("CliffordGym 32q (64 x 64)", 64, 64, 65536, 512),
                                    ("CliffordGym 20q (40 x 40)", 40, 40, 65536, 512), ("PauliGym 20q (40 x 45)", 40, 45, 8192, 512))
if os.environ.get("ONLY"):
    SHAPES = (SHAPES[int(os.environ["ONLY"])],)
NEW_ONLY = bool(os.environ.get("NEW_ONLY"))
for name, rows, cols, B, hidden in SHAPES:
    g = torch.Generator(device="cpu").manual_seed(1)
    words = torch.randint(0, 2**62, (B, rows), generator=g, dtype=torch.int64)
    if cols < 64:
        words &= (1 << cols) - 1
    words = words.cuda()
    w = (torch.randn((hidden, rows * cols), generator=g) * 0.05).to(torch.bfloat16).cuda()
    bias = torch.randn(hidden, generator=g).cuda()
    packed = pack_embed_words(w, rows, cols)
    out = torch.empty((B, hidden), dtype=torch.bfloat16, device="cuda")
    x = torch.empty((B, rows, cols), dtype=torch.bfloat16, device="cuda")
    bb = bias.to(torch.bfloat16)
    t_new = timeit(lambda: embed_words(words, cols, packed, bias, hidden, relu=True, out=out))
    if NEW_ONLY:
        print(f"{name} x {B} envs -> {hidden}: embed_words {t_new:.1f} us", flush=True)
        continue
    t_exp = timeit(lambda: expand_packed(words, cols, torch.bfloat16, out=x))
    t_gemm = timeit(lambda: _linear_relu(x.view(B, -1), w, bb))
    t_pack = timeit(lambda: pack_embed_words(w, rows, cols, out=packed))
    flop = 2.0 * B * rows * cols * hidden
    print(f"{name} x {B} envs -> {hidden}: embed_words {t_new:.1f} us ({flop / t_new / 1e6:.0f} TFLOP/s on the dense shape) | expand {t_exp:.1f} us + GEMM {t_gemm:.1f} us"
          f" | repack {t_pack:.1f} us", flush=True)
