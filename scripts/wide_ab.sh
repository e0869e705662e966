set -e
cd $GRAFT_REPO_ROOT
for o in "item_wide=0" "item_wide=1" "item_wide=1 tile_w=16"; do
  echo "== $o"
  timeout -k 10 200 python scripts/kbench.py C3 C2 $o
done
for o in "--opt item_wide=0" "--opt item_wide=1"; do
  echo "== traffic $o"
  timeout -k 10 300 bash scripts/traffic_quick.sh C3 $o
done
echo "== in flight traffic wide=0/1 (12 in flight)"
