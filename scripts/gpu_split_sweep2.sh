cd "${GRAFT_REPO_ROOT:-/root/repo}"
for kv in "" "--kv8"; do
for B in 6 12 20 24 40 48; do
  B=$B LS=1030,2048,4096,7700 VARS=101,102,103,104,105,106 NL=8 ROUNDS=3 timeout 280 python scripts/bench_attn.py $kv 2>/dev/null | grep "L=" | sed -e 's/variant //g' -e 's/ *[0-9]* GB\/s//g'
done
done
