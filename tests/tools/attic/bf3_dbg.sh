for c in 452 442 432 422 223 224 234 851; do echo "== cfg $c"; SHAPES=one CFGS=$c timeout 120 python tests/tools/bf3_bench.py 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-400; done
