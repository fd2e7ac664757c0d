# tools/launch_config_sweeps.sh -- the launch-configuration sweeps behind the defaults of late round 3, each on ONE handle at a
# time (tools/ab_same.py: options 8 = order, 2 = strip rows, 10 = tiles per period for even / odd XCDs), several handles each
echo "== G2, resident image, plain state blocks (library default)"
AB_HANDLES=3 python tools/ab_same.py "8=1,2=19" "8=1,2=10" "8=1,10=504,2=10" "8=0,2=19" "8=0,2=10" 2>&1 | grep -E "handle|M[0-9]" | cut -c1-330
echo "== G2, resident image, placement windows (CVS_PLACEMENT_SEARCH=1)"
CVS_PLACEMENT_SEARCH=1 AB_HANDLES=2 python tools/ab_same.py "8=1,2=19" "8=1,2=10" "8=1,10=504,2=10" "8=0,2=19" "8=0,2=10" 2>&1 | grep -E "handle|M[0-9]" | cut -c1-330
echo "== G2, eight rotating inputs (fresh images), plain blocks, then windows"
AB_ROT=1 AB_HANDLES=2 python tools/ab_same.py "8=0,2=10" "8=1,2=10" "8=1,10=504,2=10" "8=0,2=19" "8=1,10=504,2=19" 2>&1 | grep -E "handle|rotating" | cut -c1-330
CVS_PLACEMENT_SEARCH=1 AB_ROT=1 AB_HANDLES=2 python tools/ab_same.py "8=0,2=10" "8=1,2=10" "8=1,10=504,2=10" "8=0,2=19" "8=1,10=504,2=19" 2>&1 | grep -E "handle|rotating" | cut -c1-330
echo "== G4 pair launch"
AB_KIND=4 AB_HANDLES=2 python tools/ab_same.py "8=0,2=40" "8=0,2=27" "8=0,2=53" "8=0,2=14" "8=1,2=40" "8=1,10=504,2=40" 2>&1 | grep -E "handle|M6" | cut -c1-330
echo "== 32 x 1080p frame batch, state kept (strip rows, tiles per period, order)"
python tools/c4_config_sweep.py 5 2>&1 | grep handle
