"""tools/show_bench.py <bench line file>: the numbers of a bench.py line that the round's notes quote"""
import json, sys
j = json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
print('value', round(j['value'], 1), 'ms/step', round(j['ms_per_step'], 3), 'three-launch value', round(j['three_launch_value'] or 0, 1))
r = j['roofline']
print('roofline k_step us', round(r['avg_launch_us'], 2), 'frac', round(r['frac'], 3), 'of copy', round(r['frac_of_measured_copy'], 3), 'traffic', r.get('traffic'), 'x2', r.get('traffic_fetch_x2'))
print('  three', {k: round(v, 1) for k, v in r['three_launch']['kernels_us'].items()}, 'pipeline_frac', round(r['three_launch']['pipeline_frac'], 3))
v = j['voxelizer_only']
f = lambda x: f"{x['us_per_step']:.1f}us wall {x['k_step_us']:.1f}us kernel frac {x['kernel_frac']:.3f} | three {x['three_launch']['us_per_step']:.1f}"
print('vox B=4      ', f(v)); print('vox one sweep', f(v['one_sweep_per_launch'])); print('vox row-major', f(v['row_major_order'])); print('vox c1       ', f(v['c1_shapes']))
print('vox same buf ', round(v['same_output_buffer']['k_step_us'], 1))
ff = j['fused_feature_net']
print('fused e2e', round(ff['value'], 1), 'three-launch call us', round(ff['roofline']['us_per_call'], 1), '| pipelined e2e', round(ff['pipelined']['value'], 1), 'call us', round(ff['pipelined']['roofline']['us_per_call'], 1), 'kernel', round(ff['pipelined']['roofline']['k_step_us'], 1), 'frac', round(ff['pipelined']['roofline']['frac'], 3))
t = j['train_c3']; tt = t['targets']
print('train_c3', round(t['value'], 1), 'ms', round(t['ms_per_step'], 2), '| targets B us', round(tt['us_per_call'], 2), 'frac', round(tt['frac'], 3), 'moved frac', round(tt['frac_of_moved_bytes'], 3), '| one sample us', round(tt['one_sample_per_launch_us'], 2), 'frac', round(tt['one_sample_per_launch_frac'], 3))
s = j['stress_c5']
print('stress B', {k: round(x, 3) if x < 10 else round(x, 1) for k, x in s['pipelined'].items() if isinstance(x, (int, float))}, 'three', round(s['three_launch_us_per_step'], 1), '| one sweep kernel', round(s['one_sweep_per_launch']['k_step_us'], 1), round(s['one_sweep_per_launch']['kernel_frac'], 3))
if 'end_to_end' in s: e = s['end_to_end']; print('stress e2e', round(e['value'], 1), 'sweeps/s', round(e['ms_per_step'], 2), 'ms/step k_step', round(e['k_step_us'], 1), round(e['k_step_frac'], 3))
d = j['reference_default']
print('refdef B', round(d['voxelizer']['k_step_us'], 1), round(d['voxelizer']['kernel_frac'], 3), 'one', round(d['one_sweep_per_launch']['k_step_us'], 1), round(d['one_sweep_per_launch']['kernel_frac'], 3), 'targets one', round(d['target_assign_us'], 1), 'batch', round(d['target_assign_batch']['us_per_call'], 1), round(d['target_assign_batch']['frac'], 3))
for k, x in j.get('dropin_host', {}).items():
    if isinstance(x, dict): print('dropin', k, {a: round(x[a], 3) for a in ('hip_ms', 'cpu_ms', 'speedup')}, x['meets_50x'])
c = j['cpu_baseline']; print('cpu', round(c['value'], 1), 'c1', round(c['c1']['value'], 1), 'workers', round(c['workers']['value'], 1), 'all', round(c['all_cores']['value'], 1), c['all_cores']['processes'])
print('overlapped', round(j['overlapped']['value'], 1), round(j['overlapped']['k_step_us_while_overlapped'], 1))
