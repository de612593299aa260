#!/usr/bin/env python3
"""Micro-probe: is [hipGraph replay G1] -> [eager command E] -> [hipGraph replay G2] on ONE stream ordered when another stream replays
graphs at the same time?  G1 bumps a buffer X twenty times (20 small kernels), G2 copies X to Y through a kernel chain; after each
round Y must hold X's new value.  E in {none, memcpy, kernel, wait_event (an event recorded on the OTHER stream)}.
usage (GPU box): python tools/graph_order_probe.py"""
import torch

dev = torch.device("cuda", 0)
n = 1 << 20


def build(stream):
    X = torch.zeros(n, device=dev)
    Y = torch.zeros(n, device=dev)
    Z = torch.zeros(n, device=dev)
    W = torch.zeros(n, device=dev)
    with torch.cuda.stream(stream):
        for _ in range(3):
            X.add_(1); Y.copy_(X); W.copy_(Z)
        stream.synchronize()
        g1 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g1, stream=stream):
            for _ in range(20):
                X.add_(1)
        g2 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g2, stream=stream):
            t = X * 1
            for _ in range(5):
                t = t + 0
            Y.copy_(t)
            W.copy_(Z * 1)
    stream.synchronize()
    return X, Y, Z, W, g1, g2


def heavy(stream):
    a = torch.randn(4096, 4096, device=dev, dtype=torch.float16)
    with torch.cuda.stream(stream):
        for _ in range(3):
            b = a @ a
        stream.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=stream):
            b = a
            for _ in range(10):
                b = (b @ a) * 1e-3
    stream.synchronize()
    return g, a


def main():
    sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
    X, Y, Z, W, g1, g2 = build(sA)
    gh, keep = heavy(sB)
    for mode in ("none", "memcpy", "kernel", "wait_event", "memcpy+wait_event"):
        for concurrent in (False, True):
            bad_y = bad_w = 0
            rounds = 200
            torch.cuda.synchronize()
            for it in range(rounds):
                ev = torch.cuda.Event()
                if concurrent:
                    with torch.cuda.stream(sB):
                        gh.replay()
                        ev.record()
                        gh.replay()
                else:
                    with torch.cuda.stream(sB):
                        ev.record()
                with torch.cuda.stream(sA):
                    g1.replay()
                    if "memcpy" in mode:
                        Z.copy_(X)
                    if mode == "kernel":
                        torch.mul(X, 1, out=Z)
                    if "wait_event" in mode:
                        sA.wait_event(ev)
                    g2.replay()
                sA.synchronize()
                want = float(X[0].item())
                if not bool((Y == want).all()):
                    bad_y += 1
                if mode in ("memcpy", "kernel", "memcpy+wait_event") and not bool((W == want).all()):
                    bad_w += 1
            torch.cuda.synchronize()
            print(f"E={mode:18s} other stream busy={concurrent!s:5s}: Y stale in {bad_y}/{rounds} rounds, W (copy of E's output) stale in {bad_w}/{rounds}", flush=True)


if __name__ == "__main__":
    main()
