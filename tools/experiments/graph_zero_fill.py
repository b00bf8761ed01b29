"""Which ways of producing a zero tensor inside a HIP-graph recording replay correctly: each output is dirtied
eagerly between replays and must be zero again after every replay."""
import torch

dev = torch.device("cuda")
x = torch.randn(64, 64, device=dev)
makers = {
    "zeros_like": lambda: torch.zeros_like(x),
    "zeros": lambda: torch.zeros(64, 64, device=dev),
    "new_full": lambda: x.new_full((64, 64), 0.0),
    "empty_zero_": lambda: torch.empty_like(x).zero_(),
    "empty_fill_": lambda: torch.empty_like(x).fill_(0.0),
    "mul0": lambda: x * 0.0,
}
for name, make in makers.items():
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        make()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = make()
        out2 = out + 1.0
    bad = []
    for r in range(4):
        out.fill_(7.0)
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        if float(out.abs().max()) != 0.0 or float((out2 - 1).abs().max()) != 0.0:
            bad.append(r)
    print(f"{name:12s} wrong replays: {bad}", flush=True)
