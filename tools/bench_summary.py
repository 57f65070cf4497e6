import sys, json
for line in sys.stdin:
    line = line.rstrip()
    if line.startswith('{'):
        d = json.loads(line); r = d['roofline']
        print('B=%d value=%.0f Msamples/s ms/step=%.2f dom=%s' % (d['config']['partition'], d['value'], d['ms_per_step'], r['kernel']))
        print('   kernel_ms', {k: round(v, 3) for k, v in r['kernel_ms'].items()})
        print('   kernel_GBs', {k: round(v) for k, v in r['kernel_GBs'].items()})
        if 'cpu_baseline' in d: print('   cpu', d['cpu_baseline'])
    elif 'amdgpu.ids' not in line:
        print(line[:300])
