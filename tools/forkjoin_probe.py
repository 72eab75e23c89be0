"""How much does a fork-join of two small kernels cost inside a hipGraph, against the same two kernels one after the other?"""
import torch
dev = torch.device("cuda", 0)
a = torch.zeros(65536 * 32, dtype=torch.int32, device=dev)
b = torch.zeros(512 * 256, dtype=torch.int32, device=dev)
main, side = torch.cuda.Stream(), torch.cuda.Stream()
T = 128

def seq():
    for _ in range(T):
        a.add_(1)
        b.add_(1)

def forked():
    for _ in range(T):
        ev = torch.cuda.Event()
        ev.record(main)
        side.wait_event(ev)
        with torch.cuda.stream(side):
            b.add_(1)
            ev2 = torch.cuda.Event()
            ev2.record(side)
        a.add_(1)
        main.wait_event(ev2)

def one():
    for _ in range(T):
        a.add_(1)

for name, body in (("one kernel per pair", one), ("two kernels, one stream", seq), ("two kernels, fork-join", forked)):
    with torch.cuda.stream(main):
        body()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=main):
            body()
        torch.cuda.synchronize()
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(main)
        for _ in range(4):
            g.replay()
        e1.record(main)
        torch.cuda.synchronize()
    print(f"{name:28s}: {e0.elapsed_time(e1) * 1e3 / (4 * T):6.2f} us per pair")
