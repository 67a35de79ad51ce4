# K3 build/launch variants (run through gpurun): "<build flags>|<workgroups per CU>" ...
for v in "$@"; do
  flags="${v%%|*}"; per="${v##*|}"
  make -C slicer_amd/csrc -B EXTRA="$flags" > /dev/null 2>&1
  SLICER_K3_PER_CU=$per timeout -k 10 200 python bench.py --cpu-baseline off 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());k=d['kernels'];print('[$v]', 'dep/s %.3e'%d['value'], '%.2f ms/step'%d['ms_per_step'], {n:round(v['avg_us']) for n,v in k.items()})"
done
make -C slicer_amd/csrc -B > /dev/null 2>&1
