"""How much of the discriminator segment (graph "dfb") hides behind the generator segment (graph "g") when both are replayed at the same time on
two streams?  A TIMING probe: the values are meaningless (dfb reads what g writes); nothing here is the training step."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402


def main():
    dev = "cuda:0"
    tr = bench.build_trainer(3, 16, 256, dev, graphs=True)
    x = torch.rand(16, 3, 256, 256, device=dev) * 2 - 1
    it = 0
    for q in (2, 4):
        for _ in range(tr.graph_warmup + 2):
            it += 1
            tr.optimize_parameters(it, {"real_images": x, "rate_ind": q})
    torch.cuda.synchronize()
    G = tr.graphs._graphs
    print("graph keys:", list(G))
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

    def timed(fn, reps=10):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps * 1e3

    for q in (2, 4):
        key = ("s3", q)
        g, dfb, u, d = (G[(n, key)] for n in ("g", "dfb", "u", "d"))

        def one(gr):
            def f():
                with torch.cuda.stream(s1):
                    gr.replay()
            return f

        def both():
            with torch.cuda.stream(s1):
                g.replay()
            with torch.cuda.stream(s2):
                dfb.replay()

        def seq():
            with torch.cuda.stream(s1):
                g.replay()
                dfb.replay()
        tg, td, tu, tdd = timed(one(g)), timed(one(dfb)), timed(one(u)), timed(one(d))
        print(f"q={q}: g {tg:.2f} ms  dfb {td:.2f} ms  u {tu:.2f} ms  d {tdd:.2f} ms  sum {tg + td + tu + tdd:.2f}")
        print(f"      g then dfb {timed(seq):.2f} ms   g || dfb {timed(both):.2f} ms")


if __name__ == "__main__":
    main()
