#!/usr/bin/env python3
"""What a frame costs the HOST: a scene small enough that the GPU is never the limit (a dozen shapes on a 256 x 256 canvas), rendered
device-resident with three frame slots — frames/s = 1 / (host time per `forma_hip_render` call: argument checks, ten kernel
launches, one event record) — and with one frame in flight (+ the wait for the frame).  Also the time of the Python / ctypes
layer alone (`forma_hip_version`, a call that does nothing).  Decides whether capturing the frame in a hipGraph could pay:
a frame whose kernels take longer than the host needs to enqueue the next one is not launch-bound.

    python tools/enqueue_cost.py [--frames 3000]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=3000)
    args = ap.parse_args()
    import forma_amd
    from forma_amd import api
    import scene as S
    comp = S.random_cubics(n=12, width=256, height=256, seed=5)
    c = forma_amd.Context(0)

    from oracle import oracle as orc                       # tables only: the flatten tables of the test scenes come from the checker's builder
    o = orc.Oracle(); t = comp.tables(o); S.load(c, t)
    W = H = 256
    out = {}
    for slots in (1, 3):
        c.set_frames_in_flight(slots)
        for _ in range(3 * slots + 5):
            c.render(W, H, device_only=True)
        c.sync()
        t0 = time.perf_counter()
        for _ in range(args.frames):
            c.render(W, H, device_only=True)
        c.sync()
        out["us_per_frame_%d_slot" % slots] = round((time.perf_counter() - t0) / args.frames * 1e6, 1)
    _, tm = c.render(W, H, device_only=True, timings=True)
    out["kernels_us_of_a_frame"] = round(sum(us for _n, _s, _t0, us in c.kernel_times()), 1)
    out["kernel_launches"] = len(c.kernel_times())
    out["kernel_floors_us"] = {n: round(us, 1) for n, _s, _t0, us in c.kernel_times()}     # (what each launch costs when it has next to nothing to do)
    t0 = time.perf_counter()
    for _ in range(args.frames):
        c._L.forma_hip_version()
    out["ctypes_call_us"] = round((time.perf_counter() - t0) / args.frames * 1e6, 2)
    print(out)
    c.close()


if __name__ == "__main__":
    main()
