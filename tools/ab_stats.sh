cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 )
line() { python bench.py "$@" --steps 1000 --warmup 50 --no-extras --no-cpu-baseline 2>&1 | python -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        j = json.loads(ln); r = j['roofline']; print(j.get('ms_per_step'), 'kernel ms', r.get('launch_ms'))
"; }
for rep in 1 2 3; do
for v in r4base tree; do
  if [ $v = tree ]; then unset ADSB_LIB_PATH; else export ADSB_LIB_PATH=$PWD/adsbdec_amd/lib_ab/$v/libadsbdec_amd.so; fi
  echo "$v rep $rep plain: $(line)   stats: $(line --stats)"
done
done
unset ADSB_LIB_PATH
python tools/dense_probe.py 2>&1 | grep "dense10 {}\|gate_storm {}\|noise {}"
