for bo in "0,0,0" "0,0,2" "0,0,6" "10,10,0" "20,20,0" "30,30,0" "20,30,2" "30,40,2" "40,40,4" "10,30,1" "0,30,0" "20,0,0"; do
  echo -n "backoff $bo: "; Q1ENV_SERVER_BACKOFF=$bo python tools/time_persistent.py --envs 65536 --reps 3 2>/dev/null | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        d=json.loads(ln); print(round(d['server_us_per_tick'],3), round(d.get('server_two_streams_us_per_tick',0),3))"
done
