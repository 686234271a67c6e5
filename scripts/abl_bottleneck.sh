for d in ${BN_DBGS:-0 1 2 3 16}; do echo "== dbg=$d"; POSERISK_BN_DBG=$d timeout -k 10 120 python scripts/exp_bottleneck.py 256 2>&1 | grep "bottleneck64\|equal" | tail -2; done
