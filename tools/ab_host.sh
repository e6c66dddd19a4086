for r in 1 2 3; do for L in libmirhi_old.so libmirhi.so; do
MIRHI_LIB_NAME=$L python bench.py --no-cpu-baseline --other-workloads '' 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); a=d['rerecorded_submit']; b=d['frames_in_flight_2']['rerecorded']
print('$L', 'rerec4', a['value'], a['host_us_per_frame'], 'rerec2', b['value'], b['host_us_per_frame'])"
done; done
python tools/host_prof.py 2>&1 | tail -10
