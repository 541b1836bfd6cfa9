cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/dev
timeout 600 python -m pytest tests/test_golden_ref.py tests/test_rigid_body.py tests/test_contact_wrench_cone.py -m gpu -q 2>&1 | tail -3
(time timeout 900 python bench.py > gpurun_out/dev/bench.json 2> gpurun_out/dev/bench.err); tail -3 gpurun_out/dev/bench.err
python - <<'P'
import json
b=json.load(open('gpurun_out/dev/bench.json'))
r=b['roofline']; print({k:r[k] for k in r if 'stream' in k or 'copy' in k or k in('frac','kernel_ms')})
o=b['other_configs']
for k in ('icub_nv32_jump_N30','icub_nv35_jump_N30'):
    e=o[k]; print(k, {x:e.get(x) for x in ('sqp_newton_iteration_ms','sqp_iters_per_sec','sqp_phase_ms','sqp_status_nonzero_instances')})
    print('  ', json.dumps(e.get('roofline_condense'))[:300]); print('  ', json.dumps(e.get('closed_loop'))[:900])
print(b['sqp_iteration']['phase_ms'], b['value'])
P
