# integer vs f64 tile cells at different loads per (plane, tile) bin (run through gpurun)
for extra in "--side 256 --files 1 --snapshots 2" "--side 256 --files 1 --snapshots 2 --planes 1" "--npix 8192" "--npix 8192 --planes 1" "--npix 2048"; do for e in 0 2; do
SLICER_K4_INT=$e timeout -k 10 200 python bench.py --cpu-baseline off --parity off --e2e off $extra 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());k=d['kernels'];print('[int=$e $extra]', 'dep/s %.3e'%d['value'], '%.2f ms/step'%d['ms_per_step'], {n:round(v['avg_us']) for n,v in k.items()})"
done; done
