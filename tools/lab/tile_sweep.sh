# tile-size / slack sweep of the fused tile kernels (tools/tilebench.py; one MI355X)
for TM in 64 96 128; do for S in 1 2 4; do echo "== hiv4096 TMAX=$TM slackdiv=$S"; GCONV_TILE_NODES_MAX=$TM GCONV_TILE_SLACK_DIV=$S python tools/tilebench.py hiv4096 2>&1 | grep "hiv"; done; done
for S in 1 2 3 4 6; do echo "== hiv slackdiv=$S"; GCONV_TILE_SLACK_DIV=$S python tools/tilebench.py hiv 2>&1 | grep "hiv"; done
for T in 16 24 32; do echo "== esol T=$T"; GCONV_TILE_NODES_OVERRIDE=$T python tools/tilebench.py esol 2>&1 | grep "esol"; done
