#!/usr/bin/env python3
"""What a dependency between two kernel launches costs on this runtime, apart from any of this repo's kernels: a chain of N
tiny kernels (x += 1 on 64 elements), each waiting for the one before it — on ONE stream, or alternating between the caller's
stream and a second one, tied by events (two SIDE streams waiting for each other's events fault hipStreamEndCapture on ROCm
7.2, see capture_probe.py) — issued launch by launch and replayed from a hipGraph.  Prints the time per link."""
import time, torch
N = 40
x = torch.zeros(64, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
evs = [torch.cuda.Event() for _ in range(N + 1)]
def chain_one_stream():
    for _ in range(N):
        x.add_(1.0)
def chain_two_streams():
    cur = torch.cuda.current_stream()
    evs[0].record(cur)
    for k in range(N):
        s = s1 if k % 2 == 0 else cur
        s.wait_event(evs[k])
        with torch.cuda.stream(s):
            x.add_(1.0)
            evs[k + 1].record(s)
    cur.wait_event(evs[N])
def timed(f, reps=50):
    for _ in range(5): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps
def graphed(f):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        f(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            f()
    return g
for name, f in (("one stream", chain_one_stream), ("two streams, an event per link", chain_two_streams)):
    e = timed(f)
    g = graphed(f)
    r = timed(g.replay)
    print(f"{name:32s}: launch by launch {e / N * 1e6:6.2f} us per link, replayed from a hipGraph {r / N * 1e6:6.2f} us per link")
