for v in 2 3 4; do
  echo "== wpe $v"; FTL_LIB=$PWD/variants_wpe$v.so python3 bench.py --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel_ms'])"
done
