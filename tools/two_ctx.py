"""How much throughput do k contexts (one HIP stream each, one host thread each) rendering the same scene concurrently
give on one GPU?  usage: python tools/two_ctx.py [workload] [frames]"""
import sys, os, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from forma_amd import api, scenes
wl = sys.argv[1] if len(sys.argv) > 1 else "paris-like-30k-4k"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 60
build, W, H = scenes.WORKLOADS[wl]
comp = build()
def make():
    r = api.Renderer(0)
    img = np.zeros(W * H * 4, np.uint8)
    r.render(comp, api.BufferBuilder(img, api.LinearLayout(W, W * 4, H)).build(), api.RGBA, api.Color(1, 1, 1, 1), None)
    for _ in range(3):
        r._ctx.render(W, H, clear=(1, 1, 1, 1), device_only=True)
    return r
rs = [make() for _ in range(3)]
for k in (1, 2, 3):
    def work(r, n):
        for _ in range(n):
            r._ctx.render(W, H, clear=(1, 1, 1, 1), device_only=True)
    ths = [threading.Thread(target=work, args=(rs[i], frames // k)) for i in range(k)]
    t0 = time.perf_counter()
    for t in ths: t.start()
    for t in ths: t.join()
    dt = time.perf_counter() - t0
    print(f"{k} context(s): {(frames // k) * k / dt:8.1f} frames/s")
