# Kernel-pass size: the same snapshot in 8, 4, 2, 1 sub-files (run through gpurun)
for f in 8 4 2 1; do
  timeout -k 10 200 python bench.py --cpu-baseline off --parity off --e2e off --files $f 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());k=d['kernels'];print('[files $f]', 'dep/s %.3e'%d['value'], '%.2f ms/step'%d['ms_per_step'], {n:[round(x) if isinstance(x,float) else x for x in v.values()] for n,v in k.items()})"
done
