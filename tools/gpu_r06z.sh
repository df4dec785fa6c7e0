#!/bin/bash
# round 6: the stand-alone variable-base multiexps after the heavy buckets went onto rows (and the G2 thresholds): launch lists, the lone proof, the micro line
export TMPDIR=/tmp
for g in g1 g2; do
  OUT=gpurun_out/r06z_vb_$g; mkdir -p $OUT
  timeout 900 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o t -- python tools/vb_trace.py run $g > $OUT/run.txt 2>&1
  f=$(find $OUT/trace -name '*kernel_trace.csv' | head -1)
  echo "== $g"; python tools/vb_trace.py read "$f" | tee $OUT/launch_list.txt | grep -v "rocclr\|k_msm_task_[obh]"
  find $OUT/trace -type f -size +1M -delete
done
echo "-- lone proof"; python tools/lone_probe.py 2>&1 | grep -v amdgpu | tail -2
python bench.py --micro-only 2>/dev/null | tail -1 > gpurun_out/r06z_micro.json
python3 -c "
import json
d=json.loads(open('gpurun_out/r06z_micro.json').read())
m=d.get('micro',d)
for k,v in m.items():
    if isinstance(v,dict): print(k, {kk:v[kk] for kk in v if kk in ('ms','mscalar_per_s','accumulate_share','one_shot_ms','kernel_ms')})
"
