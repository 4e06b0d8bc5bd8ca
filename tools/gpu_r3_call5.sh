#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r3_call5; rm -rf $OUT; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
(time python bench.py) > $OUT/default.json 2> $OUT/default.err; tail -3 $OUT/default.err
python - <<PY
import json
d=json.loads(open("$OUT/default.json").read().strip().splitlines()[-1])
r=d["roofline"]
print(d["value"], d["ms_per_step"], "frac", r["frac"], r["mean_launch_ms"], r["rays_per_launch"], "alone", r["alone"]["frac"], "traffic", r["traffic"], "lane util", r["valu_lane_util"])
print(d.get("self_check")); print(json.dumps(d.get("cpu_baseline"))[:300])
print({k:(round(v["mean_launch_ms_alone"],3), round(v["mean_launch_ms_overlapped"] or 0,3), round(v["valu_lane_util"],3), round(v["launches_per_sample"],2)) for k,v in (r["kernels"] or {}).items()})
PY
python - <<PY
import sys; sys.path[:0]=['.','tests']
import torch, scenes, time
from redner_amd import redner as rd, trim_cache
from redner_amd.render_pytorch import RenderFunction
dev=torch.device('cuda:0')
sc=scenes.bunny_box(dev, resolution=(1024,1024))
args=RenderFunction.serialize_scene(sc, 8, 4, sampler_type=rd.SamplerType.sobol, device=dev, backend=rd)
img=RenderFunction.apply(1,*args); img.sum().backward(); torch.cuda.synchronize()
print('parked after a 1024x1024 8-spp forward+backward: %.2f GB' % (trim_cache()/2**30))
PY
