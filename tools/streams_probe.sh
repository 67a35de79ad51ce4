# Snapshots in flight per GPU (run through gpurun)
for k in 1 2 3; do
  timeout -k 10 200 python bench.py --cpu-baseline off --parity off --e2e off --streams $k --steps 16 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());k=d['kernels'];print('[streams $k]', 'dep/s %.3e'%d['value'], '%.2f ms/step'%d['ms_per_step'], {n:[round(x) if isinstance(x,float) else x for x in v.values()] for n,v in k.items()})"
done
