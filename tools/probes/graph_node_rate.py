"""How fast does a captured HIP graph issue DEPENDENT kernel nodes (r05)?  A frame-sharded forward at 2 frames per rank is
~1000 launches of 5-40 us: if a graph node costs several us to issue, launch count - not kernel time - bounds the 8-GPU step.
Chains of N tiny / small kernels on one stream and on two forked streams inside one graph, replayed.
usage (GPU box): python tools/probes/graph_node_rate.py"""
import time

import torch


def chain(n, x, size):
    for _ in range(n):
        x = x[:size].add_(1.0) if size else x.add_(1.0)
    return x


def run(n, two, numel):
    dev = "cuda"
    a, b = torch.zeros(numel, device=dev), torch.zeros(numel, device=dev)
    side, other = torch.cuda.Stream(), torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        chain(3, a, 0)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
        main = torch.cuda.current_stream()
        if two:
            other.wait_stream(main)
            with torch.cuda.stream(other):
                chain(n, b, 0)
        chain(n, a, 0)
        if two:
            main.wait_stream(other)
    g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        g.replay()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / reps


def run_two_graphs(n, numel):
    """the two chains as TWO graphs, each captured on its own stream and replayed there, joined by events"""
    dev = "cuda"
    a, b = torch.zeros(numel, device=dev), torch.zeros(numel, device=dev)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    graphs = []
    for st, x in ((s1, a), (s2, b)):
        st.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(st):
            chain(3, x, 0)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st, capture_error_mode="thread_local"):
            chain(n, x, 0)
        graphs.append(g)

    def step():
        cur = torch.cuda.current_stream()
        for st, g in zip((s1, s2), graphs):
            st.wait_stream(cur)
            with torch.cuda.stream(st):
                g.replay()
        cur.wait_stream(s1)
        cur.wait_stream(s2)

    step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        step()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / reps


def main():
    print(f"{'kernel size':>14s} {'nodes/chain':>11s} {'one chain ms':>13s} {'us/node':>8s} {'forked graph ms':>16s} {'ratio':>6s} "
          f"{'two graphs ms':>14s} {'ratio':>6s}")
    for numel in (1024, 1 << 20, 8 << 20):
        for n in (250, 1000):
            t1, t2, t3 = run(n, False, numel), run(n, True, numel), run_two_graphs(n, numel)
            print(f"{numel * 4 / 1024:11.0f} KiB {n:11d} {t1:13.2f} {1e3 * t1 / n:8.2f} {t2:16.2f} {t2 / t1:6.2f} {t3:14.2f} {t3 / t1:6.2f}")


if __name__ == "__main__":
    main()
