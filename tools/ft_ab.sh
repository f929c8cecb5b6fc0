#!/bin/bash
# A/B timing of the fine-tune step over library builds (tools/ab_bench.py --build-only makes them):   bash tools/ft_ab.sh base ftnt512 ...
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
cd "$R"; export MULUT_NO_BUILD=1; mkdir -p gpurun_out/ft_ab
for v in "$@"; do
  if [ "$v" = base ]; then unset MULUT_LIB; else export MULUT_LIB=$R/build/variants/libmulut_$v.so; fi
  timeout -k 10 200 python bench.py --config 4 > gpurun_out/ft_ab/$v.json 2> gpurun_out/ft_ab/$v.err || echo "$v failed"
  python - "$v" <<'PY'
import json, sys
v = sys.argv[1]
try:
    d = json.loads(open("gpurun_out/ft_ab/%s.json" % v).read().strip().splitlines()[-1])
    print(v, "ms_per_step", d["ms_per_step"])
except Exception as e:
    print(v, "no result", e)
PY
done
