#!/bin/bash
# MFMA conv kernels with vector row loads (operand-layout copies): parity + A/B
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2d
(AIMNET_SPLIT_MAX=0 timeout 900 python -m pytest tests -q -x -m gpu -k "not fuzz" 2>&1 | tail -5) > gpurun_out/r2d/tests_mfma_all.txt
for m in 0 1 2 3; do
  AIMNET_CONV_MFMA=$m python bench.py --no-cpu-baseline --steps 30 2>/dev/null | tail -1 > gpurun_out/r2d/bench_mfma$m.json
done
python - <<'PY'
import json
for m in range(4):
    try:
        d = json.load(open(f"gpurun_out/r2d/bench_mfma{m}.json"))
        print(m, round(d["ms_per_step"], 4), {k: round(v, 3) for k, v in d["family_ms_per_step"].items()})
    except Exception as e:
        print(m, "failed", e)
PY
tail -4 gpurun_out/r2d/tests_mfma_all.txt
