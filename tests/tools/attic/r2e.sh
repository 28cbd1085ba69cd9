#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2e
(AIMNET_SPLIT_MAX=0 AIMNET_CONV_MFMA=1 timeout 900 python -m pytest tests -q -x -m gpu -k "parity or calculator" 2>&1 | tail -3) > gpurun_out/r2e/tests_mfma_fwd.txt
(AIMNET_CONV_MFMA=0 timeout 900 python -m pytest tests -q -x -m gpu -k "not fuzz" 2>&1 | tail -3) > gpurun_out/r2e/tests_valu.txt
for m in 0 1 2; do
  AIMNET_CONV_MFMA=$m python bench.py --no-cpu-baseline --steps 30 2>/dev/null | tail -1 > gpurun_out/r2e/bench_mfma$m.json
done
AIMNET_CONV_MFMA=0 AIMNET_HIP_LIB=gpurun_in/bwd3.so python bench.py --no-cpu-baseline --steps 30 2>/dev/null | tail -1 > gpurun_out/r2e/bench_mfma9.json
python - <<'PY'
import json
for m in (0, 1, 2, 9):
    try:
        d = json.load(open(f"gpurun_out/r2e/bench_mfma{m}.json"))
        print(m, round(d["ms_per_step"], 4), {k: round(v, 3) for k, v in d["family_ms_per_step"].items()})
    except Exception as e:
        print(m, "failed", e)
PY
tail -2 gpurun_out/r2e/tests_mfma_fwd.txt gpurun_out/r2e/tests_valu.txt
