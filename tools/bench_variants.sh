# bench.py across the variants quoted in DESIGN.md (one line each)
for extra in "" "--planes 1" "--clustered" "--hydro" "--accum f64" "--accum fixed64" "--mas ngp" "--algo direct --steps 2 --warmup 1" "--npix 1024 --side 256 --files 1 --snapshots 2" "--npix 16384 --planes 1" "--npix 8192"; do
  timeout -k 10 200 python bench.py --cpu-baseline off --parity off --e2e off $extra 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());k=d['kernels'];print('[%s]'%'$extra', 'dep/s %.3e'%d['value'], 'in/s %.3e'%d['n_in_per_s'], '%.2f ms/step'%d['ms_per_step'], {n:round(v['avg_us']) for n,v in k.items()})"
done
