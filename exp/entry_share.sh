# the entry search with four lanes per beam against one lane per beam, by stage, for 1..8 ranks' shares at both resolutions
for R in 3840x2160 7680x4320; do
  for E in 0 1; do
    if [ $E = 1 ]; then export RT_ENTRY_ONE_LANE=1; else unset RT_ENTRY_ONE_LANE; fi
    echo "== $R one_lane=$E"
    PYTHONPATH=. timeout -k 10 300 python3 -u profiles/tile_share_stages.py $R || exit 1
  done
done
