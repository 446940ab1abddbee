for i in 1 2; do
  echo "old:"; EINX_LIB=ab_libs/libeinx_oldnms.so python bench.py --no-cpu-baseline --no-extras --no-scale-legs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
  echo "new:"; python bench.py --no-cpu-baseline --no-extras --no-scale-legs 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done
echo old; EINX_LIB=ab_libs/libeinx_oldnms.so python tools/nms_bench.py | tail -1
echo new; python tools/nms_bench.py | tail -1
echo old b1; EINX_LIB=ab_libs/libeinx_oldnms.so python tools/latency_graph.py 2>&1 | tail -3
echo new b1; python tools/latency_graph.py 2>&1 | tail -3
