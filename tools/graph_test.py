"""hipGraph replay vs eager forward (run on the GPU box: python tools/graph_test.py)."""
import os, sys, time


def main():
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, ROOT)
    import torch, bench
    from fdn_hip.pipeline import GraphedForward, forward_streams
    dev = torch.device("cuda:0")
    net, lp = bench.build_models(dev)
    for (B, h, w) in [(1, 256, 256), (1, 720, 1280), (8, 720, 1280)]:
        x = bench.make_input(B, h, w, dev, 1)
        ref = forward_streams(net, lp, x, 1)
        gf = GraphedForward(net, lp)
        out = gf(x); torch.cuda.synchronize()
        print((B, h, w), "graph == eager:", torch.equal(out, ref))
        for name, fn in (("eager", lambda: forward_streams(net, lp, x, 1)), ("graph", lambda: gf(x))):
            fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
            n = 10 if B == 1 else 3
            for _ in range(n): fn()
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
            print(f"   {name}: {dt*1e3:.2f} ms/call  {B/dt:.2f} img/s", flush=True)


if __name__ == "__main__":
    main()
