# Host time per phase of the fenced frame loop, two builds interleaved on one box: libmirhi_old.so = the commit before (git stash; python renderer-rs_amd/build.py --variant old; git stash pop)
# against libmirhi.so; then tools/host_prof.py (needs build.py --variant hp -DMIRHI_HOST_PROF).
for r in 1 2 3; do for L in libmirhi_old.so libmirhi.so; do
MIRHI_LIB_NAME=$L python bench.py --no-cpu-baseline --other-workloads '' 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); a=d['rerecorded_submit']; b=d['frames_in_flight_2']['rerecorded']
print('$L', 'rerec4', a['value'], a['host_us_per_frame'], 'rerec2', b['value'], b['host_us_per_frame'])"
done; done
python tools/host_prof.py 2>&1 | tail -10
