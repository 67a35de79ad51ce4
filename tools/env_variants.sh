# bench.py under environment-variable variants (run through gpurun): "VAR=val VAR2=val" ...
for v in "$@"; do
  env $v timeout -k 10 200 python bench.py --cpu-baseline off 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());k=d['kernels'];print('[$v]', 'dep/s %.3e'%d['value'], '%.2f ms/step'%d['ms_per_step'], {n:round(v['avg_us']) for n,v in k.items()})"
done
