import json,sys
for f in sys.argv[1:]:
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1])
    except Exception as e:
        print(f, 'ERR', e); continue
    print(f, 'value=%.0f a-s/s  ms/step=%.3f  kernel_ms=%s' % (d['value'], d['ms_per_step'], d.get('kernel_time_ms_per_step')))
    for k,v in d.get('kernels',{}).items():
        print('   %-26s n=%4d avg_us=%9.2f ms/step=%7.3f  %s %s frac=%s' % (k, v['launches'], v['avg_us'], v['ms_per_step'], v.get('achieved'), v.get('unit'), v.get('frac')))
    if 'cpu_baseline' in d: print('   cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
