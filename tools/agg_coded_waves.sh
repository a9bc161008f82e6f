rows=${1:-600}
for env in "QSX_AGG_JIT_WAVES=0" "QSX_AGG_JIT_WAVES=5" "QSX_AGG_JIT_WAVES=6" "QSX_AGG_JIT_WAVES=5 QSX_AGG_ACC_KIB=8"; do
  echo "== $env"
  env $env QSX_DEBUG_LAUNCH=1 timeout -s KILL 120 python3 tools/agg_coded_probe.py $rows 2> /tmp/coded_ab.err | python3 -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print({k:(round(v,3) if isinstance(v,float) else v) for k,v in d.items() if 'ms' in k or 'same' in k})"
  grep "jit launch\|shape launch" /tmp/coded_ab.err | sort | uniq -c | head -4
done
