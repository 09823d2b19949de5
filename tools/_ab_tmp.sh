python -m pytest tests/test_gpu_eigen.py tests/test_gpu_parity.py tests/test_gpu_parity_portable_libm.py tests/test_analytic.py -x -q -m gpu > gpurun_out/ab_par.log 2>&1; tail -3 gpurun_out/ab_par.log
for r in 1 2; do for d in 1 0; do DES_E2_DEFER=$d python bench.py --steps 400 --warmup 40 --cpu-steps 0 --no-ceiling 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['config'].get('kernel_ms_per_call',{})
print('defer $d', '%.2f us' % (1e3*d['ms_per_step']), {a[:6]: round(1e3*b,1) for a,b in k.items() if a[:2] in ('E2',)})"
done; done
python tools/time_yield.py 2>/dev/null
