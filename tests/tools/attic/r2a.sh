#!/bin/bash
# round 2, first look at the MFMA conv kernels: layout probe, parity of both kernel families, kernel timings
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2a
python -m pytest tests/test_gpu_ops.py -q -x -m gpu -k "mfma" 2>&1 | tail -15 > gpurun_out/r2a/probe.txt
(AIMNET_SPLIT_MAX=0 timeout 900 python -m pytest tests -q -x -m gpu -k "not fuzz" 2>&1 | tail -25) > gpurun_out/r2a/tests_mfma_all.txt
(AIMNET_SPLIT_MAX=0 AIMNET_CONV_MFMA=0 timeout 900 python -m pytest tests -q -x -m gpu -k "parity or calculator" 2>&1 | tail -8) > gpurun_out/r2a/tests_valu_all.txt
for m in 0 1 2 3; do
  AIMNET_CONV_MFMA=$m python bench.py --no-cpu-baseline --steps 30 2>/dev/null | tail -1 > gpurun_out/r2a/bench_mfma$m.json
done
python - <<'PY'
import json
for m in range(4):
    try:
        d = json.load(open(f"gpurun_out/r2a/bench_mfma{m}.json"))
        print(m, round(d["ms_per_step"], 4), {k: round(v, 3) for k, v in d["family_ms_per_step"].items()})
    except Exception as e:
        print(m, "failed", e)
PY
cat gpurun_out/r2a/probe.txt; tail -12 gpurun_out/r2a/tests_mfma_all.txt; tail -5 gpurun_out/r2a/tests_valu_all.txt
