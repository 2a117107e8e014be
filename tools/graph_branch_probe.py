#!/usr/bin/env python3
"""Do two independent branches of a captured hipGraph run side by side?  Two one-workgroup spin kernels (torch.cuda._sleep)
on two streams, eager and as one graph: time of both / time of one."""
import torch


def timed(fn, reps=5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    cyc = 2_000_000
    side = torch.cuda.Stream()

    def one():
        torch.cuda._sleep(cyc)

    def two():
        cur = torch.cuda.current_stream()
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            torch.cuda._sleep(cyc)
        torch.cuda._sleep(cyc)
        cur.wait_stream(side)

    print("eager: one %.3f ms, two branches %.3f ms" % (timed(one), timed(two)))
    g1, g2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
    with torch.cuda.graph(g1):
        one()
    with torch.cuda.graph(g2):
        two()
    print("graph: one %.3f ms, two branches %.3f ms" % (timed(g1.replay), timed(g2.replay)))


if __name__ == "__main__":
    main()
