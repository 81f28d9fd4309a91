export CPC_ENABLE_PROBES=1
run() { env $2 python bench.py --steps 40 --warmup 20 --no-cpu-baseline --no-secondary --no-trainer-loop --no-score-gemm 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$1', d['ms_per_step'], d['ms_per_step_median'])"; }
for rep in 1 2; do
run default CPC_NOOP=1
run no_prep "CPC_PROBE_SKIP=cpc_conv_w_prep,cpc_cast2d,cpc_prep_frag"
run no_reduce "CPC_PROBE_SKIP=cpc_reduce_conv_w,cpc_reduce_slabs,cpc_colsum"
done
