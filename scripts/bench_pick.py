import json,sys
for f in sys.argv[1:]:
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        g=d.get('graph_replay',{})
        print(f, 'two-in-flight', round(d['ms_per_step'],3), 'single', round(d.get('single_in_flight_ms_per_step',0),3), 'eager', round(d.get('eager_ms_per_step',0),3), 'enc replay', g.get('encode',{}).get('median_ms'), 'enc+q', g.get('encode_plus_query',{}).get('median_ms'), 'equal', d.get('pipelined_outputs_equal_eager'))
    except Exception as e: print(f, 'ERR', e)
