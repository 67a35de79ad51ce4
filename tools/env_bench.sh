# bench under environment-variable variants: bash tools/env_bench.sh "VAR=1 VAR2=x" "..." [-- extra bench args]
args=""
vars=()
for a in "$@"; do if [ "$a" = "--" ]; then shift; args="$*"; break; fi; vars+=("$a"); shift; done
for v in "${vars[@]}"; do
  env $v timeout -k 10 200 python bench.py --cpu-baseline off --parity off --e2e off $args 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read());k=d['kernels'];print('[$v $args]', 'dep/s %.3e'%d['value'], '%.3f ms/step'%d['ms_per_step'], {n:round(v['avg_us'],1) for n,v in k.items()})"
done
