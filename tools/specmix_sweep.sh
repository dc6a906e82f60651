export MTD_LAB=1
for cfg in "0 0 2" "2 2 2" "4 4 2" "6 6 2" "8 8 2" "2 2 3" "4 4 3" "3 3 5" "1 1 5" "2 0 5"; do
  set -- $cfg
  MTD_SPECMIX_STAGGER=$1 MTD_SPECMIX_STAGGER_BWD=$2 MTD_SPECMIX_STAGGER_MOD=$3 timeout -k 10 120 python tools/specmix_probe.py 2>&1 | tail -1
done
MTD_SPECMIX_COLS=2 timeout -k 10 120 python tools/specmix_probe.py 2>&1 | tail -1
