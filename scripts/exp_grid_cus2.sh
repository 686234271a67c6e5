cd "${GRAFT_REPO_ROOT:-$(pwd)}"
run() {
  echo "== $*"
  A="$1"; shift
  env "$@" python3 bench.py $A --no-other-configs --no-roofline --cpu-frames 0 --steps 100 --warmup 10 --repeats 3 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('value', d['value'], 'spread', d['value_spread']['min'], d['value_spread']['max'])"
}
for rep in 1 2; do
run "--precision bf16 --batch 256 --lanes 2" X=0
for C in 112 144 160 176 208; do
run "--precision bf16 --batch 256 --lanes 3" POSERISK_GRID_CUS=$C
done
run "--precision bf16 --batch 256 --lanes 2" POSERISK_GRID_CUS=224
done
