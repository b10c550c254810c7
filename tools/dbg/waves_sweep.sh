#!/bin/bash
for w in 2 4 8 12 16; do
  TOKSCORE_WAVES=$w timeout -k 10 200 python bench.py --no-cpu --steps 10 > gpurun_out/sw_$w.json 2>/dev/null
  python - <<PY
import json
d=json.load(open("gpurun_out/sw_$w.json"))
ms=d["roofline"]["kernel_ms"]; lanes=256*$w*64
iters=1.155e8/lanes
print("waves $w: scan %.3f ms, %.0f lookups per lane, %.0f cycles per lookup step (2.4 GHz)" % (ms, iters, ms*1e-3*2.4e9/iters))
PY
done
