# final tree of round 3: full GPU suite, smoke, the default bench line
OUT=gpurun_out/r03_final3; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; tail -4 $OUT/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?"; python -c "
import json; d=json.load(open('$OUT/bench.json')); r=d['reduced_precision']
print(d['value'], d['roofline']['frac'], d['roofline']['gemm']['tflops'], r['value'], r['roofline']['achieved'], r['roofline']['gemm']['tflops'], d['parity_vs_reference_golden']['final_cloud_max_abs'], d['cpu_baseline']['value'])"
