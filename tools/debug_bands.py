"""Dev tool: the bench's config-2 frame as one context vs as two row bands (same process, sequential)."""
import hashlib, os, sys
import numpy as np
import torch
torch.cuda.init()
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ray_tracer_webgl_amd import abi, dist as ptdist, scenes
from ray_tracer_webgl_amd.tracer import PathTracer

def render(rank, world, use_torch, reserve):
    sc = scenes.config2(1920, 1080, 16, 64, 50)
    p = sc.params.copy()
    p.band_rows, p.band_index, p.band_count = ptdist.band_of(rank, world, 4)
    p.time_step = abi.PT_TIME_STEP_DECORRELATED
    pt = PathTracer(1920, 1080, use_torch=use_torch)
    pt.set_spheres(sc.spheres); pt.set_params(p); pt.reserve_passes(reserve); pt.tune(8)
    q = p.copy(); q.time = 0.0; q.first_pass = 0
    pt.set_params(q); pt.render_passes(64)
    a = pt.accum().copy()
    seg = pt.stats().segments
    pt.close()
    return a, seg

full, seg1 = render(0, 1, False, 64)
print("single", hashlib.sha256(full.tobytes()).hexdigest(), seg1)
for use_torch, reserve in ((False, 64), (True, 64), (True, 128)):
    parts = [render(r, 2, use_torch, reserve) for r in range(2)]
    out = np.zeros_like(full)
    for r in range(2):
        ys = np.asarray(abi.owned_rows(1080, 4, r, 2))
        out[ys] = parts[r][0][: len(ys)]
    bad = np.nonzero((out.view(np.uint32) != full.view(np.uint32)).any(axis=(1, 2)))[0]
    print("bands torch=%s reserve=%d:" % (use_torch, reserve), hashlib.sha256(out.tobytes()).hexdigest(), parts[0][1] + parts[1][1],
          "rows differing:", len(bad), bad[:16])
    if len(bad):
        y = bad[0]
        xs = np.nonzero((out[y].view(np.uint32) != full[y].view(np.uint32)).any(axis=1))[0]
        print("  row", y, "pixels differing", len(xs), xs[:8], out[y, xs[0]], full[y, xs[0]])

# the frame bench.py --gpus 2 --backend gloo --same-device hashed, if it was dumped (PT_BENCH_DUMP)
dump = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "bench2_frame.npy")
if os.path.exists(dump):
    b = np.load(dump)
    bad = np.nonzero((b.view(np.uint32) != full.view(np.uint32)).any(axis=(1, 2)))[0]
    print("bench 2-rank frame:", hashlib.sha256(b.tobytes()).hexdigest(), "rows differing:", len(bad), bad[:24])
    if len(bad):
        y = bad[0]
        xs = np.nonzero((b[y].view(np.uint32) != full[y].view(np.uint32)).any(axis=1))[0]
        print("  row", y, "pixels differing", len(xs), xs[:8], b[y, xs[0]], full[y, xs[0]])
        print("  rows mod 8:", np.bincount(bad % 8, minlength=8))
