# bench_iter with and without an environment switch, interleaved (dev tool): ab_env.sh VAR [bench_iter args]
v=$1; shift
for r in 1 2 3; do
  echo "off: $(python3 tools/bench_iter.py "$@" 2>&1 | grep chunk)"
  echo "on : $(env $v=1 python3 tools/bench_iter.py "$@" 2>&1 | grep chunk)"
done
