"""Records the SHA-256 of the full config-2 frame produced by the HIP path (run on a GPU box):

    python tests/golden/make_full_digests.py

The digest pins the WHOLE 1920x1080x1024-spp frame across rounds; its correctness rests on the
oracle window spot-checks and the partition / additivity properties of
tests/test_gpu_properties.py, which hold for the same frame."""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from ray_tracer_webgl_amd import scenes  # noqa: E402
from ray_tracer_webgl_amd.tracer import render_scene  # noqa: E402

from ray_tracer_webgl_amd import abi  # noqa: E402

sc = scenes.config2()
t, a = render_scene(sc)
out = {"config2_1920x1080_1024spp": {
    "sha256": hashlib.sha256(np.ascontiguousarray(a, dtype=np.float32).tobytes()).hexdigest(),
    "segments": int(t.stats().segments)}}
t.close()
sc = scenes.config2(1920, 1080, 16, 64, 50)  # the pass shape bench.py renders
sc.params.time_step = abi.PT_TIME_STEP_DECORRELATED
t, a = render_scene(sc)
out["config2_1920x1080_64x16spp_decorrelated"] = {
    "sha256": hashlib.sha256(np.ascontiguousarray(a, dtype=np.float32).tobytes()).hexdigest(),
    "segments": int(t.stats().segments)}
t.close()
# BASELINE config 3 as bench.py --config 3 renders it: 3840x2160, 4096 spp = 256 passes of 16 spp with
# decorrelated pass times, 64 passes per launch (pass-order accumulation: any launch grouping gives the same bits)
sc = scenes.config3(3840, 2160, 16, 256, 50)
sc.params.time_step = abi.PT_TIME_STEP_DECORRELATED
t, a = render_scene(sc, passes_per_launch=64)
out["config3_3840x2160_256x16spp_decorrelated"] = {
    "sha256": hashlib.sha256(np.ascontiguousarray(a, dtype=np.float32).tobytes()).hexdigest(),
    "segments": int(t.stats().segments)}
t.close()
# the stress configs as bench.py --config 4 / --config 5 render them (16-spp passes, decorrelated pass times)
for key, sc, ppl in (("config4_1024x1024_512x16spp_decorrelated", scenes.config4(1024, 1024, 16, 512, 50), 64),
                     ("config5_1920x1080_16x16spp_decorrelated", scenes.config5(1920, 1080, 16, 16, 50), 16)):
    sc.params.time_step = abi.PT_TIME_STEP_DECORRELATED
    t, a = render_scene(sc, passes_per_launch=ppl)
    out[key] = {"sha256": hashlib.sha256(np.ascontiguousarray(a, dtype=np.float32).tobytes()).hexdigest(),
                "segments": int(t.stats().segments)}
    t.close()
path = os.path.join(HERE, "full_frame_digests.json")
if os.environ.get("GRAFT_REPO_ROOT"):
    path = os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "full_frame_digests.json")
json.dump(out, open(path, "w"), indent=1)
print(json.dumps(out))
