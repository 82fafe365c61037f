"""Dev tool: would two overlapping half-launches shorten a short launch's drain?  One rank's band
of eight (config 2, 64 passes of 16 spp) rendered (a) as one launch, (b) as two 32-pass launches
of two contexts on two streams (the second kernel's workgroups start as the first's exit)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ray_tracer_webgl_amd import abi, dist as ptdist, scenes
from ray_tracer_webgl_amd.tracer import PathTracer

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
passes, spp = 64, 16
sc = scenes.config2(1920, 1080, spp, passes, 50)
p = sc.params.copy()
p.band_rows, p.band_index, p.band_count = ptdist.band_of(0, n, 4)
p.time_step = abi.PT_TIME_STEP_DECORRELATED

def make(stream, n_passes, first):
    with torch.cuda.stream(stream):
        pt = PathTracer(1920, 1080, use_torch=True)
        pt.set_spheres(sc.spheres)
        q = p.copy(); q.first_pass = first
        pt.set_params(q)
        pt.reserve_passes(n_passes)
        pt.set_geometry_path(abi.PT_GEOM_GRID)
    return pt

s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
one = make(s0, passes, 0)
halves = [make(s0, passes // 2, 0), make(s1, passes // 2, passes // 2)]
quarters = [make(s0 if k % 2 == 0 else s1, passes // 4, k * passes // 4) for k in range(4)]

def run(ctxs, streams, counts):
    ts = []
    for rep in range(5):
        for c in ctxs: c.reset()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for c, st, k in zip(ctxs, streams, counts):
            with torch.cuda.stream(st):
                c.render_passes(k)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    return min(ts[1:])

a = run([one], [s0], [passes])
b = run(halves, [s0, s1], [passes // 2] * 2)
c = run(quarters, [s0, s1, s0, s1], [passes // 4] * 4)
print("rank 0 of %d: one launch %.2f ms | two overlapping half-launches %.2f ms | four quarter-launches on two streams %.2f ms" % (n, a, b, c))
