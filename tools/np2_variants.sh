# Build-time variants on a 4000^2 map (run through gpurun): bash tools/np2_variants.sh "<flags>" ...
set -e
for v in "$@"; do
  make -C slicer_amd/csrc -B EXTRA="$v" > /dev/null 2>&1
  for extra in "--npix 4000" "--npix 4000 --planes 1"; do
  timeout -k 10 200 python bench.py --cpu-baseline off --parity off --e2e off $extra 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());k=d['kernels'];print('[$v $extra]', 'dep/s %.3e'%d['value'], '%.2f ms/step'%d['ms_per_step'], {n:round(v['avg_us']) for n,v in k.items()})"
  done
done
make -C slicer_amd/csrc -B > /dev/null 2>&1
