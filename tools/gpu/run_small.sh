python bench.py --steps 300 --warmup 2 --no-cpu-baseline > /tmp/b.json 2>/dev/null &
BP=$!
while kill -0 $BP 2>/dev/null; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|Package Power" | sed 's/.*: //' | tr '\n' ' '; echo
  sleep 2
done | sort | uniq -c | sort -k1,1n | tail -12
tail -c 4000 /tmp/b.json | grep -o '"value": [0-9.]*' | head -1
