#!/bin/bash
# nfa:: kernel times of a short bench run (run on the GPU box from the repo root):  tools/kernel_times.sh [extra bench args]
export TMPDIR=/tmp
D=$(mktemp -d /tmp/ktXXXX)
rocprofv3 --kernel-trace --stats --output-format csv -d $D -o bench -- python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-profile --no-other-mode "$@" > $D/line.json 2> $D/err.txt
python - <<PY
import json
d = json.loads(open("$D/line.json").read().strip().splitlines()[-1])
print("ms_per_step", d["ms_per_step"], "rays/iter", d["config"].get("rays_per_iter_timed"), "samples/s", d.get("samples_per_sec"))
PY
python tools/kernel_summary.py $D | grep "nfa::"
rm -rf $D
