for extra in "--clustered" "--planes 1" "--npix 1024 --side 256 --files 1 --snapshots 1" "--mas ngp" "--accum fixed64" "--accum f64"; do
  timeout -k 10 120 python bench.py --cpu-baseline off --steps 4 --warmup 1 $extra 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());k=d['kernels'];print('$extra', '%.3e'%d['value'], 'in %.3e'%d['n_in_per_s'], '%.2f ms'%d['ms_per_step'], {n:round(v['avg_us']) for n,v in k.items()})"
done
